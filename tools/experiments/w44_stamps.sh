#!/bin/bash
# build host: D=face-detection-and-tracking_amd/csrc; touch $D/conv_wino44.h; make -C $D -j8 EXTRA=-DFDT_W44_STAMPS; cp $D/libfdt_hip.so tools/experiments/w44_libs/libfdt_hip_stamps.so; touch $D/conv_wino44.h; make -C $D -j8
mkdir -p gpurun_out/r4stamps
FDT_LIB=$PWD/tools/experiments/w44_libs/libfdt_hip_stamps.so python tools/experiments/w44_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4stamps/w44_stamps.txt
