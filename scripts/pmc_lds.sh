# PMC passes on the LDS path of the C2 kernel: bytes moved, stalls, average latency of an LDS instruction
set -e
TAG=${1:-lds}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d gpurun_out/$TAG/pmc_a -o p -- python bench.py --steps 1 --warmup 0 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads --option fused_partner=0 > gpurun_out/$TAG/pmc_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc LdsLatency LdsUtil --output-format csv -d gpurun_out/$TAG/pmc_b -o p -- python bench.py --steps 1 --warmup 0 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads --option fused_partner=0 > gpurun_out/$TAG/pmc_b.log 2>&1
