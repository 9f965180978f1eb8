# rocprofv3 evidence for profiles/: kernel trace + stats, PMC passes (each in its own run), HBM traffic.
set -e
TAG=${1:-r03}
# BENCH_ARGS: extra bench.py arguments, e.g. "--workload c5"; the PMC passes add --option fused_partner=0 (rocprofv3
# serialises kernels while it collects counters, so the main instance must do the whole frame in every launch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -o t -- python bench.py --steps 3 --warmup 1 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads $BENCH_ARGS > gpurun_out/$TAG/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/$TAG/pmc_a -o p -- python bench.py --steps 1 --warmup 0 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads --option fused_partner=0 $BENCH_ARGS > gpurun_out/$TAG/pmc_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG/pmc_b -o p -- python bench.py --steps 1 --warmup 0 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads --option fused_partner=0 $BENCH_ARGS > gpurun_out/$TAG/pmc_b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/pmc_fetch -o p -- python bench.py --steps 1 --warmup 0 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads --option fused_partner=0 $BENCH_ARGS > gpurun_out/$TAG/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/pmc_write -o p -- python bench.py --steps 1 --warmup 0 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads --option fused_partner=0 $BENCH_ARGS > gpurun_out/$TAG/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$TAG/pmc_l2 -o p -- python bench.py --steps 1 --warmup 0 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads --option fused_partner=0 $BENCH_ARGS > gpurun_out/$TAG/pmc_l2.log 2>&1
find gpurun_out/$TAG -name "*.csv" | head -20
