#!/bin/bash
# times the struct-stage backward of every ablation build under tools/bin (diagnostic; results of those builds are wrong by design)
export STAGE_ONLY2=1
for v in "$@"; do
  echo "== ablation $v"
  MGV_ALLOW_ABLATION=1 MGV_LIB=$PWD/tools/bin/libabl$v.so python tools/bench_stage.py 64 5 2>&1 | grep -v amdgpu.ids || exit 1
done
