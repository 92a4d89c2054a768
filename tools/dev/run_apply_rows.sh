# chunk heights of K2 (TM_APPLY_ROWS) in the single-sweep relaxation pass
for r in 18 6 3 4 5 8 12; do echo "TM_APPLY_ROWS=$r"; STEADY_SINGLE=1 TM_APPLY_ROWS=$r python3 tools/dev/steady_time.py 4096 2048 1024; done
