#!/bin/bash
# reference scripts/run_ocr.sh on the MI355X hot path: per-frame OCRNet (resnet101dilated + ocrnet_deepsup)
ARCH=res101_ocrnet; CFGNAME=vsp-resnet101dilated-ocr_deepsup.yaml
source "$(dirname "$0")/_frame_job.sh"
