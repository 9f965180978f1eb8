# PMC passes on the HBM-resident workload (C5 final scene): L2 hit rate of the BVH / primitive fetches, HBM bytes,
# VALU active-lane fraction
set -e
TAG=${1:-c5}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$TAG/pmc_l2 -o p -- python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/$TAG/pmc_l2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/pmc_fetch -o p -- python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/$TAG/pmc_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/pmc_write -o p -- python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/$TAG/pmc_write.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/$TAG/pmc_a -o p -- python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/$TAG/pmc_a.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/$TAG/pmc_b -o p -- python bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/$TAG/pmc_b.log 2>&1
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -o t -- python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/trace.log 2>&1
