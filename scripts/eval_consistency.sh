#!/bin/bash
# VC_n and TC of a prediction dump (reference VC_perclip.py / TC_cal.py): scripts/eval_consistency.sh <VSPW_480p> <pred dir>
PKG=cvpr2021_vspw_implement_amd
python -m $PKG.VC_perclip --dataroot $1 --pred $2 --split val.txt --clip_num 16
python -m $PKG.TC_cal --dataroot $1 --pred $2 --split val.txt
