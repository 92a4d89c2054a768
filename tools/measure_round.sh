#!/bin/bash
# The probes behind the numbers DESIGN.md / README.md quote, in one run on the MI355X box (through gpurun, from the repo root):
#   tools/measure_round.sh r02   ->  gpurun_out/<tag>_measurements.txt   (copy into profiles/)
tag=${1:-r04}
out=gpurun_out/${tag}_measurements.txt
mkdir -p gpurun_out
{
  echo "# $tag measurements, $(date -u +%Y-%m-%dT%H:%MZ), $(python3 -c 'import torch; print(torch.cuda.get_device_name(0))' 2>/dev/null)"
  echo "## K1 (tools/tfi_probe.py)"; python3 tools/tfi_probe.py 4096 2>/dev/null; python3 tools/tfi_probe.py 2048 2>/dev/null
  echo "## single-rank relaxation sweeps at settled clocks (tools/dev/steady_time.py): three per pass (fixed perimeter), then two per pass"
  python3 tools/dev/steady_time.py 1024 1448 2048 2896 4096 2>/dev/null
  TM_FUSE_3=0 python3 tools/dev/steady_time.py 1024 1448 2048 2896 4096 2>/dev/null
  echo "## the same from a cold start, consecutive calls of 200 sweeps (tools/dev/ramp_probe.py)"
  python3 tools/dev/ramp_probe.py 4096 200 12 2>/dev/null
  TM_FUSE_3=0 python3 tools/dev/ramp_probe.py 4096 200 12 2>/dev/null
  echo "## shader clock / power while the pass runs (tools/dev/clock_probe.py)"
  python3 tools/dev/clock_probe.py 3 2>/dev/null | tail -4 | cut -c1-400
  echo "## STREAM-style copy variants, 256 MiB and 64 MiB per array, alternating direction (tools/ubench/stream.hip)"
  [ -x tools/ubench/stream_bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/stream_bin tools/ubench/stream.hip 2>/dev/null
  if [ -x tools/ubench/stream_bin ]; then ./tools/ubench/stream_bin 256 1 | grep -E "MiB|grid  *8192|hipMemcpy"; ./tools/ubench/stream_bin 64 1 | grep -E "MiB|grid  *8192|hipMemcpy"; fi
  echo "## one rank of a strip, transport that moves nothing (tools/split_path_cost.py), signal / wait kernels"
  for n in 4096 2048 1024; do python3 tools/split_path_cost.py $n 2>/dev/null | grep Native; done
  echo "## the same with hipEvent record / wait (TM_PAIR_SYNC=events)"
  for n in 4096 2048; do TM_PAIR_SYNC=events python3 tools/split_path_cost.py $n 2>/dev/null | grep Native; done
  echo "## BASELINE configs[3] / [4] on one GPU"; python3 tools/config4_probe.py 2>/dev/null | tail -2; python3 tools/config5_probe.py 2>/dev/null | tail -2
  echo "## the reference's example inputs as written, 10 Picard iterations, hip solver (tools/t106_probe.py)"
  python3 tools/t106_probe.py T106 2>/dev/null | tail -1; python3 tools/t106_probe.py LS89 2>/dev/null | tail -1
  echo "## plain BiCGStab (diagonal), time per inner iteration (tools/bicgstab_iter_probe.py)"
  python3 tools/bicgstab_iter_probe.py 4096 200 2>/dev/null | tail -1; python3 tools/bicgstab_iter_probe.py 1024 500 2>/dev/null | tail -1; python3 tools/bicgstab_iter_probe.py 256 500 2>/dev/null | tail -1
  echo "## perturbed 4096^2 block to a scaled residual <= 1e-8 (tools/solve_probe.py)"; python3 tools/solve_probe.py 4096 1e-6 1e-6 1e-10 2>/dev/null
  echo "## the same with one operator application per pass in the multigrid cycle (TM_MG_PAIR=0)"; TM_MG_PAIR=0 python3 tools/solve_probe.py 4096 1e-6 1e-6 1e-10 2>/dev/null
  echo "## host-buffer seam, PCIe inclusive (tools/oneshot_probe.py)"; python3 tools/oneshot_probe.py 2>/dev/null | tail -4
  echo "## cross-queue ordering by mechanism + the co-residency self-test (tools/ubench/queue_order.hip), then the same with GPU_MAX_HW_QUEUES=1"
  [ -x tools/ubench/queue_order_bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/ubench/queue_order_bin tools/ubench/queue_order.hip 2>/dev/null
  if [ -x tools/ubench/queue_order_bin ]; then timeout -k 10 120 ./tools/ubench/queue_order_bin 2000 | grep -v "^done"; GPU_MAX_HW_QUEUES=1 timeout -k 10 120 ./tools/ubench/queue_order_bin 300 | grep -E "GPU_MAX|self-test"; fi
  echo "## BiCGStab iteration by size: 64-column tiles with halo loads / overlapping 62-column strips (tools/dev/vk_overlap_sizes.py)"
  python3 tools/dev/vk_overlap_sizes.py 1024 1448 2048 2896 4096 2>/dev/null
  echo "## GMRES(30) on the device against BiCGStab: T106 as written, 10 Picard iterations (tools/t106_probe.py T106 gmres)"
  python3 tools/t106_probe.py T106 gmres 2>/dev/null | tail -1
} > "$out" 2>&1
cat "$out"
