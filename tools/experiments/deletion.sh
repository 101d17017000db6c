#!/bin/bash
# What is each layer group worth to the multi-stream step?  FDT_SKIP_OPS (csrc/model.hip, experiment hook) leaves ops out of
# the launch sequence by name prefix; results are wrong by construction, only the step time counts.
#   bash tools/experiments/deletion.sh "--height 480 --width 640"      (or "" for 1024x1024)
# the hooks this script sets are compiled in only with -DFDT_EXPERIMENTS (the product library ignores them):
(cd face-detection-and-tracking_amd/csrc && touch model.hip conv.hip && make -s -j8 EXTRA=-DFDT_EXPERIMENTS > /dev/null)
SZ="$1"
SSH1=conv2_SSH.conv1,conv3_SSH.conv1,conv4_SSH.conv1,conv5_SSH.conv1,conv6_SSH.conv1,conv7_SSH.conv1
CTX=conv2_SSH.conv2,conv3_SSH.conv2,conv4_SSH.conv2,conv5_SSH.conv2,conv6_SSH.conv2,conv7_SSH.conv2
run() {
  FDT_SKIP_OPS="$2" python bench.py --steps 96 --warmup 12 --cpu-frames 0 --host-frames 0 --profile-frames 1 $SZ 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %8.1f frames/s  %.3f ms/step' % ('$1', d['value'], d['ms_per_step']))"
}
run "nothing deleted" ""
run "heads of levels 2-5" face_loc.2,face_loc.3,face_loc.4,face_loc.5
run "all heads" face_loc,head_finalize
run "8^2-32^2 tail" face_loc.2,face_loc.3,face_loc.4,face_loc.5,conv5_SSH.conv2,conv6_SSH,conv7_SSH,layer6,latlayer_c7
run "smooth_c3/4/5" smooth_c
run "*_SSH.conv1" $SSH1
run "SSH context convs" $CTX
run "LFPN 1x1" latlayer_,conv3_ct_py,conv4_ct_py,conv5_ct_py
run "layer1" layer1.
run "layer2" layer2.
run "layer3" layer3.
run "layer4" layer4.
run "layer1-4" layer1.,layer2.,layer3.,layer4.
