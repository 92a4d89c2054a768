# A/B of K2x3 chunk heights on one rank of a strip with a transport that moves nothing (tools/split_path_cost.py)
for r in 0 36 38 26 32; do echo "rows $r"; TM_FUSE3_ROWS=$r python3 tools/split_path_cost.py 4096 2>/dev/null | grep Native; done
for r in 0 21 26 38 14 20; do echo "rows $r"; TM_FUSE3_ROWS=$r python3 tools/split_path_cost.py 2048 2>/dev/null | grep Native; done
