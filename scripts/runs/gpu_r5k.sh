#!/bin/bash
O=gpurun_out/r5k; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$O/pmc_a -- python3 $R/scripts/pmc_lds.py > $R/$O/a.log 2>&1; echo "a rc=$?"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_b -- python3 $R/scripts/pmc_lds.py > $R/$O/b.log 2>&1; echo "b rc=$?"
cd $R
python scripts/pmc_lds.py collect $O/pmc_a $O/pmc_b | tee $O/pmc_lds.json | head -80
tail -3 $O/a.log; tail -3 $O/b.log
