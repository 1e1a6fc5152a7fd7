#!/bin/bash
# usage (GPU box, through gpurun): tools/gt.sh <tag> [pytest args]   — the -m gpu suite, summary into gpurun_out/<tag>/gputests.log
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -m gpu -x -q "$@" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -30 > gpurun_out/$tag/gputests.log
tail -6 gpurun_out/$tag/gputests.log
