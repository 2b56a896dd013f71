# Shared body of the clip-level jobs (reference scripts/run_temporal_psp.sh, run_temporal_ocr.sh, run_netwarp.sh: same
# hyper-parameters, same train -> test(val) -> test(test) sequence).  The caller sets METHOD, CLIPNUM, GPU_NUM.
# One process per GPU (RCCL) replaces the reference's single process with nn.DataParallel over --gpu_num devices.
DATAROOT=${DATAROOT:-"your/path/to/VSPW_480p"}
SAVE=${SAVE:-"./savemodel"}
PKG=cvpr2021_vspw_implement_amd
BATCHSIZE=8; WORKERS=12; CROPSIZE=479; START_GPU=0; TRAINFPS=1; EPOCH=120; LR=0.002; VAL=False
DILATION=0; DILATION2="3,6,9"; CLIPOCR_ALL=False; USEMEMORY=True; MAXDIST='3'
ALLSUP=True; ALLSUPSCALE=0.5; LINEAR_COM=True; DISTSOFTMAX=False; DISTNEAREST=False; TEMP=0.05; EARLYFUSE=True
ARCH=resnet101
CFG="$(python -c "import $PKG, os; print(os.path.dirname($PKG.__file__))")/config/vsp-${ARCH}dilated-ppm_deepsup_clip.yaml"
PRE_ENC=${PRE_ENC:-"./imgnetpre/${ARCH}-imagenet.pth"}
NAME="job_lr${LR}_bs${BATCHSIZE}_epoch${EPOCH}_clipnum${CLIPNUM}_arch${ARCH}_method${METHOD}_USEMEMORY${USEMEMORY}"
SAVEROOT=$SAVE/$NAME
echo 'train...'
python -m torch.distributed.run --nnodes=1 --nproc-per-node $GPU_NUM --master-addr 127.0.0.1 -m $PKG.train_clip2 \
  --cfg $CFG --batchsize $BATCHSIZE --workers $WORKERS --start_gpu $START_GPU --gpu_num $GPU_NUM --dataroot $DATAROOT \
  --trainfps $TRAINFPS --lr $LR --multi_scale True --saveroot $SAVEROOT --totalepoch $EPOCH --cropsize $CROPSIZE \
  --validation $VAL --clip_num $CLIPNUM --dilation_num $DILATION --earlyfuse $EARLYFUSE --allsup $ALLSUP \
  --allsup_scale $ALLSUPSCALE --linear_combine $LINEAR_COM --distsoftmax $DISTSOFTMAX --distnearest $DISTNEAREST \
  --temp $TEMP --pre_enc $PRE_ENC --max_distances $MAXDIST --method $METHOD --dilation2 $DILATION2 \
  --clipocr_all $CLIPOCR_ALL --use_memory $USEMEMORY
LOAD=$SAVEROOT/model_epoch_$EPOCH.pth
for SPLIT in val test; do
  echo "$SPLIT..."
  python -m $PKG.test_clip2 --cfg $CFG --start_gpu $START_GPU --dataroot $DATAROOT --saveroot ./saveimg/${NAME}_$SPLIT \
    --batchsize 1 --is_save True --clip_num $CLIPNUM --dilation_num $DILATION --load $LOAD --split $SPLIT \
    --allsup $ALLSUP --allsup_scale $ALLSUPSCALE --linear_combine $LINEAR_COM --distsoftmax $DISTSOFTMAX \
    --distnearest $DISTNEAREST --temp $TEMP --max_distances $MAXDIST --gpu_num 1 --method $METHOD \
    --dilation2 $DILATION2 --clipocr_all $CLIPOCR_ALL --use_memory $USEMEMORY
done
