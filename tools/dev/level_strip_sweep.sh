for w in 28 60 124 252; do
  for n in 2048 1448; do
    b=$(TM_LEVEL_STRIP=$w TM_NULL_EXCHANGE_US=10 python3 tools/split_path_cost.py $n 2>/dev/null | grep "world 3 rank 1 Native" | sed 's/.*Hooks: //; s/ us per.*//')
    echo "strip $w, $n^2, 10 us exchange: $b us per sweep"
  done
done
