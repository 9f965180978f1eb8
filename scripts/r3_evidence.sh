#!/bin/bash
# round-3 evidence: profiles for C2 / C5 / C4, split-frame probes, the full bench line, the 2-rank rehearsal
set -e
cd $GRAFT_REPO_ROOT
bash scripts/gpu_profile.sh r03_c2 > gpurun_out/prof_c2.log 2>&1
BENCH_ARGS="--workload c5" bash scripts/gpu_profile.sh r03_c5 > gpurun_out/prof_c5.log 2>&1
BENCH_ARGS="--workload c4" bash scripts/gpu_profile.sh r03_c4 > gpurun_out/prof_c4.log 2>&1
rm -f gpurun_out/r3_probes.log
for st in 1 2 4 8; do
  echo "== stride $st" >> gpurun_out/r3_probes.log
  timeout -k 10 120 python scripts/frame_probe.py --stride=$st >> gpurun_out/r3_probes.log 2>/dev/null
  timeout -k 10 120 python scripts/frame_probe.py --stride=$st --batch=8 >> gpurun_out/r3_probes.log 2>/dev/null
done
timeout -k 10 120 python scripts/fused_stats.py 100 --stats > gpurun_out/r3_stats_c2.log 2>&1
timeout -k 10 600 python bench.py > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rehearse --steps 3 --warmup 1 --no-cpu-baseline --no-workloads --no-strong-workloads --no-c5-full > gpurun_out/r3_rehearse.json 2> gpurun_out/r3_rehearse.err
echo done
# afterwards, in the container: python scripts/summarize_profile.py r03_c2 c2; ... r03_c5 c5; ... r03_c4 c4  (the second
# argument names the pmc_<workload>_latest.json that bench.py reads), commit profiles/, then re-run bench.py for the line
