cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python bench.py --workload c5 --even-axes --steps 10 --warmup 3 2>/dev/null
C5="bench.py --workload c5 --steps 3 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY -d gpurun_out/r02/c5_sq --output-format csv -- python3 $C5 > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d gpurun_out/r02/c5_sq2 --output-format csv -- python3 $C5 > /dev/null 2>&1; echo rc=$?
NDI_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats -d gpurun_out/r02/roctx --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-gather-leg --placement-probe 0 > /dev/null 2>&1; echo rc=$?
ls gpurun_out/r02/roctx/*/ | head; head -5 gpurun_out/r02/roctx/*/*marker*stats*.csv 2>/dev/null | head -20
