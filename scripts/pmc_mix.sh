# VALU instruction mix of the render kernel (two PMC passes of 8 SQ counters each)
set -e
TAG=${1:-mix}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 --output-format csv -d gpurun_out/$TAG/mix_a -o p -- python bench.py --steps 1 --warmup 0 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads --option fused_partner=0 > gpurun_out/$TAG/mix_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/$TAG/mix_b -o p -- python bench.py --steps 1 --warmup 0 --lean-run --no-cpu-baseline --no-workloads --no-strong-workloads --option fused_partner=0 > gpurun_out/$TAG/mix_b.log 2>&1
