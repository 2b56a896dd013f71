#!/bin/bash
# reference scripts/run_temporal_psp.sh on the MI355X hot path (METHOD=clip_psp, clips of 4 frames, 4 GPUs)
METHOD=clip_psp; CLIPNUM=4; GPU_NUM=${GPU_NUM:-4}
source "$(dirname "$0")/_clip_job.sh"
