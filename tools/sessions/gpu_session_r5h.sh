# GPU session r5h: SQ counters of the delivery kernels with the lean expansion (one pass of config 3 with 10 % v5 per counter group)
set -u
O=$PWD/gpurun_out/r5h
mkdir -p $O
export TMPDIR=/tmp
export RGR_DELIVER_LEAN=1
CGROUPS="SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_LDS;SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE;SQ_ACTIVE_INST_VMEM,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_MISC,SQ_INST_CYCLES_VMEM_RD,SQ_INST_CYCLES_SMEM,SQ_LDS_ADDR_CONFLICT,SQ_LDS_ATOMIC_RETURN"
timeout 900 python tools/pmc_kernels.py --out $O/pmc_deliver_kernels_lean.json --match dedup,expand \
  --groups "$CGROUPS" \
  -- python bench.py --time-format deliver --steps 1 --warmup 0 > $O/pmc_deliver_kernels_lean.txt 2> $O/pmc_deliver_kernels_lean.err
echo "pmc rc=$?"; cat $O/pmc_deliver_kernels_lean.txt | cut -c1-1200; grep "group" $O/pmc_deliver_kernels_lean.err | cut -c1-300
