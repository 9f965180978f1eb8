set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/parity3.log 2>&1 || (tail -30 gpurun_out/parity3.log; exit 1)
tail -3 gpurun_out/parity3.log
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -o r2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc2a -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc2a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc2b -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc2b.log 2>&1
ls gpurun_out/prof2 gpurun_out/pmc2a gpurun_out/pmc2b
