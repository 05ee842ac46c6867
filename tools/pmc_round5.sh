#!/bin/bash
# Round-5 hardware counters of launches AS THEY SHIP: tools/one_launch.py lowers the bench network with the shipped
# tuning table and replays ONE of its launches 8 times; the counters below are those of the LAST 8 dispatches of the named kernel —
# the plan's own descriptor, tile hint and fused epilogue (the hint is printed beside each entry).  rocprofv3 --pmc, counters only,
# separate passes; FETCH_SIZE x 2 per the guide (gfx950 reports half of wide streaming reads).
# usage (GPU box): bash tools/pmc_round5.sh  -> gpurun_out/pmc_${PMC_TAG:-r05}/{counters.json,summary.txt} (copy to profiles/<tag>_pmc/; round 6: PMC_TAG=r06)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_${PMC_TAG:-r05}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag program pattern kernel-substring
  tag=$1; prog=$2; pat=$3; kern=$4
  echo "$tag|$kern" >> $OUT/entries.txt
  python $R/tools/one_launch.py $prog "$pat" 1 2>/dev/null | grep ONE_LAUNCH > $OUT/${tag}.info
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE -d $OUT/${tag}_a --output-format csv -- python $R/tools/one_launch.py $prog "$pat" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d $OUT/${tag}_b --output-format csv -- python $R/tools/one_launch.py $prog "$pat" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${tag}_c --output-format csv -- python $R/tools/one_launch.py $prog "$pat" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/${tag}_d --output-format csv -- python $R/tools/one_launch.py $prog "$pat" > /dev/null 2>&1
}
rm -f $OUT/entries.txt
run wgrad_l3conv1            bwd "wgrad:backbone.encoder.layer3.1.conv1"   "wgrad_kernel<"
run wgrad_l4conv1_shortK     bwd "wgrad:backbone.encoder.layer4.1.conv1"   "wgrad_kernel<"
run wgrad_l4conv3_shortK     bwd "wgrad:backbone.encoder.layer4.1.conv3"   "wgrad_kernel<"
run wgrad_l3ds_shortK        bwd "wgrad:backbone.encoder.layer3.0.downsample.0" "wgrad_kernel<"
run winowg_head              bwd "wgrad:att_reg_box.2.0"                   "wino_wgrad_kernel"
run wino_head_fwd            fwd "att_reg_box.2.0"                         "wino_kernel"
run wino_P3_2                fwd "backbone.fpn.P3_2"                       "wino_kernel"
run wino_l3conv2_bnstat      fwd "backbone.encoder.layer3.1.conv2"         "wino_kernel"
run igemm_l3conv1_bnstat     fwd "backbone.encoder.layer3.1.conv1"         "igemm_kernel"
run igemm_l3conv3_bnstat     fwd "backbone.encoder.layer3.1.conv3"         "igemm_kernel"
run igemm_l3conv1_dgrad_bnb  bwd "dgrad:backbone.encoder.layer3.1.conv1"   "igemm_kernel"
run mx_stem_fwd              fwd "backbone.encoder.conv1"                  "mx_kernel"
run igemm_l1conv1_bnpre      fwd "backbone.encoder.layer1.1.conv1"         "igemm_kernel"
run pw_l1conv3_fwd           fwd "backbone.encoder.layer1.1.conv3"         "pw_kernel"
run pw_l1conv1_dgrad         bwd "dgrad:backbone.encoder.layer1.1.conv1"   "pw_kernel"
run bn_apply_l1bn3           fwd "backbone.encoder.layer1.1.bn3"           "bn_apply_kernel"
run bn_apply_l3bn2           fwd "backbone.encoder.layer3.1.bn2"           "bn_apply_kernel"
run bn_bwd_apply_l1bn3       bwd "bnbwd:backbone.encoder.layer1.1.bn3"     "bn_bwd_apply_kernel"
run bn_bwd_apply_l3bn2       bwd "bnbwd:backbone.encoder.layer3.1.bn2"     "bn_bwd_apply_kernel"
# round 6: the launches that changed — stream-K forward candidates (whatever the table picked: the hint is printed), the data gradient that
# stores the masked dout, and the apply pass behind it (no mask read, one output)
run igemm_l4conv1_fwd        fwd "backbone.encoder.layer4.1.conv1"         "igemm_kernel"
run igemm_P5_1_fwd           fwd "backbone.fpn.P5_1"                       "igemm_kernel"
run wino_l4conv2_fwd         fwd "backbone.encoder.layer4.1.conv2"         "wino_kernel"
run igemm_l2conv1_dgrad_mask bwd "dgrad:backbone.encoder.layer2.2.conv1"   "igemm_kernel"
run bn_bwd_apply_l2bn3       bwd "bnbwd:backbone.encoder.layer2.1.bn3"     "bn_bwd_apply_kernel"
cd $R && python tools/pmc_summarize.py $OUT 8 | tee $OUT/summary.txt
