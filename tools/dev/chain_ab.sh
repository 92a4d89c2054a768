# One rank of a strip (rank 1 of 3, null transport), us per sweep, with the exchange's device time emulated ON the chain (TM_NULL_EXCHANGE_US):
# the three level passes of a coupled triple as three launches against the fused level kernel.  usage: bash tools/dev/chain_ab.sh [sizes...]
for n in ${@:-2048 1448 4096 1024}; do
  for us in 0 5 10 20; do
    a=$(TM_NULL_EXCHANGE_US=$us TM_LEVELS_FUSED=0 python3 tools/split_path_cost.py $n 2>/dev/null | grep "world 3 rank 1 Native" | sed 's/.*Hooks: //; s/ us per.*//')
    b=$(TM_NULL_EXCHANGE_US=$us TM_LEVELS_FUSED=1 python3 tools/split_path_cost.py $n 2>/dev/null | grep "world 3 rank 1 Native" | sed 's/.*Hooks: //; s/ us per.*//')
    echo "$n^2 per rank, exchange ${us} us on the chain: three level launches $a us per sweep, fused level kernel $b us per sweep"
  done
done
