#!/bin/bash
# reference scripts/run_temporal_ocr.sh on the MI355X hot path (METHOD=clip_ocr, clips of 4 frames, 4 GPUs)
METHOD=clip_ocr; CLIPNUM=4; GPU_NUM=${GPU_NUM:-4}
source "$(dirname "$0")/_clip_job.sh"
