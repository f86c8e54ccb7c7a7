#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
P="rocprofv3 --kernel-trace --output-format csv"
timeout -k 10 400 $P --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/w -o p -- python3 tools/bench_gconv.py --batch 32 --only up > $O/w.log 2>&1 || { tail $O/w.log; exit 1; }
timeout -k 10 400 $P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/v -o p -- python3 tools/bench_gconv.py --batch 32 --only up > $O/v.log 2>&1 || { tail $O/v.log; exit 1; }
python3 tools/pmc_sq.py gpurun_out/r05_upconv_wait_breakdown.json $O/w $O/v > $O/sum.log 2>&1
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05_upconv_wait_breakdown.json"))
for k in d["kernels"][:9]:
    wc=k.get("SQ_WAVE_CYCLES",1)
    print(k["kernel"][:52].ljust(52), "ms %.2f"%(k["duration_us"]/1e3), "mfma %.3f"%k.get("mfma_busy",0), "parked %.2f issue-stall %.2f active %.2f lds-stall %.3f"%(k.get("SQ_WAIT_ANY",0)/wc, k.get("SQ_WAIT_INST_ANY",0)/wc, k.get("SQ_ACTIVE_INST_ANY",0)/wc, k.get("SQ_WAIT_INST_LDS",0)/wc),
          "bank-conf %.3f"%(k.get("SQ_LDS_BANK_CONFLICT",0)/max(k.get("SQ_LDS_IDX_ACTIVE",1),1)), "valu/mfma %.2f"%k.get("valu_per_mfma_cycle64",0), "salu", int(k.get("SQ_INSTS_SALU",0))>>20, "M")
PY
