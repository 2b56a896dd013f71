#!/bin/bash
# reference scripts/run_netwarp.sh on the MI355X hot path (METHOD=netwarp, clips of 2 frames, 2 GPUs)
METHOD=netwarp; CLIPNUM=2; GPU_NUM=${GPU_NUM:-2}
source "$(dirname "$0")/_clip_job.sh"
