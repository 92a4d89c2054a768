# one process, coupled blocks (strip of nb blocks of n^2): coupled sweep triples (two queues, fused level kernel) against sweep pairs, by size
for cfg in "8 128" "8 256" "8 362" "8 512" "2 512" "2 724" "4 512"; do
  set -- $cfg
  t=$(TM_TRIPLES_SINGLE_MIN_NODES=1 python3 tools/config4_probe.py $1 $2 2>/dev/null | grep "single_sweep=False" | sed 's/.*False: //; s/ us per.*//')
  p=$(TM_TRIPLES_SINGLE_MIN_NODES=999999999999 python3 tools/config4_probe.py $1 $2 2>/dev/null | grep "single_sweep=False" | sed 's/.*False: //; s/ us per.*//')
  echo "$1 x $2^2 in one process: triples $t us per sweep, pairs $p us per sweep"
done
