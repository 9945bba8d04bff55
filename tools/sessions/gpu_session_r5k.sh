# GPU session r5k: SQ counters of the compact expansions (ids24 lane-held, packed tile-per-block) and of the plain tuple kernel — are they
# bound by instruction issue like the delivery expansion was (r5c / r5h)?
set -u
O=$PWD/gpurun_out/r5k
mkdir -p $O
export TMPDIR=/tmp
CGROUPS="SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_LDS;SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE"
timeout 900 python tools/pmc_kernels.py --out $O/pmc_compact_kernels.json --match expand \
  --groups "$CGROUPS" \
  -- python bench.py --time-format ids24,packed,tuple --steps 1 --warmup 0 > $O/pmc_compact_kernels.txt 2> $O/pmc_compact_kernels.err
echo "pmc rc=$?"; cat $O/pmc_compact_kernels.txt | cut -c1-1200; grep "group" $O/pmc_compact_kernels.err | cut -c1-300
