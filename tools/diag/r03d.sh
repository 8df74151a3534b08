#!/bin/bash
out=gpurun_out/r03d; mkdir -p $out
V=$PWD/gaussian-splatting-lightning_amd/variants
tools/ab_kernel.sh 2 acc3 abl -- --no-renderer-only > $out/ab.txt 2>&1; cat $out/ab.txt
for lib in abl acc3; do
  GSPL_HIP_LIB=$V/libgspl_hip_$lib.so tools/pmc_quick.sh "" composite_bwd SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY > $out/pmc_${lib}_a.txt 2>&1
  GSPL_HIP_LIB=$V/libgspl_hip_$lib.so tools/pmc_quick.sh "" composite_bwd SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY > $out/pmc_${lib}_b.txt 2>&1
  cat $out/pmc_${lib}_a.txt $out/pmc_${lib}_b.txt
done
GSPL_BWD_KERNEL=2 tools/pmc_quick.sh "" composite_bwd SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY > $out/pmc_bwd2_a.txt 2>&1; cat $out/pmc_bwd2_a.txt
