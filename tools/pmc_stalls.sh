#!/bin/bash
# Where do the fused-loader convolutions wait?  Ten rocprofv3 --pmc passes (counters only, one block group per pass) over
# tools/conv3p_time.py (level-0 / level-1 shapes of config 2; DS_CONV_PC / DS_CONV_VEC as exported by the caller), reduced by
# tools/pmc_stalls.py.   tools/pmc_stalls.sh TAG   -> gpurun_out/r04/stalls_TAG/
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r04/stalls_$1; [ -z "$PASSES" ] && rm -rf $O; mkdir -p $O
P=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
 "TA_TA_BUSY TA_TOTAL_WAVEFRONTS"
 "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
 "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES"
 "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY"
 "TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES"
 "TCC_REQ TCC_HIT TCC_MISS TCC_TAG_STALL"
 "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL"
)
for i in ${PASSES:-"${!P[@]}"}; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc ${P[$i]} --output-format csv -d $O/p$i -o c -- python3 tools/conv3p_time.py 10 > $O/p$i.log 2>&1
  rc=$?; echo "pass $i rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out: stopping"; break; fi
done
python3 tools/pmc_stalls.py $O | tee $O/summary.txt
