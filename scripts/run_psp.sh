#!/bin/bash
# reference scripts/run_psp.sh on the MI355X hot path: per-frame PSPNet (resnet101dilated + ppm_deepsup), BASELINE cfg 2
ARCH=res101_ppm; CFGNAME=vsp-resnet101dilated-ppm_deepsup.yaml
source "$(dirname "$0")/_frame_job.sh"
