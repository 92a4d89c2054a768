# sweep triples against sweep pairs across ranks by block size (rank 1 of 3, null transport with the exchange emulated on the chain): where should LocalPlan::triple_halo start?
for us in 0 10; do
for n in ${@:-256 384 512 724 1024}; do
  t=$(TM_NULL_EXCHANGE_US=$us TM_TRIPLES_MIN_NODES=1 python3 tools/split_path_cost.py $n 2>/dev/null | grep "world 3 rank 1 Native" | sed 's/.*Hooks: //; s/ us per.*//')
  p=$(TM_NULL_EXCHANGE_US=$us TM_TRIPLES_MIN_NODES=999999999 python3 tools/split_path_cost.py $n 2>/dev/null | grep "world 3 rank 1 Native" | sed 's/.*Hooks: //; s/ us per.*//')
  echo "$n^2 per rank, exchange $us us: triples $t us per sweep, pairs $p us per sweep"
done
done
