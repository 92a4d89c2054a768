# kernel timeline of the multi-rank sweep triples on one rank (null transport): usage bash tools/dev/run_trace_split.sh [n]
n=${1:-2048}
root=$(pwd); out=$root/gpurun_out/trace_split_$n; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python3 $root/tools/split_path_trace.py $n 60 > $out/run.out 2> $out/run.err
cd $root
python3 tools/trace_tail.py $(find $out -name p_kernel_trace.csv | head -1) 120 $out/tail.csv
