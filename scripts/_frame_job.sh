# Shared body of the per-frame jobs (reference scripts/run_psp.sh, run_ocr.sh: train.py -> test.py on val and test).
# The caller sets ARCH and CFGNAME.
DATAROOT=${DATAROOT:-"/your/path/to/LVSP_plus_data_label124_480p"}
SAVE=${SAVE:-"./savemodel"}
PKG=cvpr2021_vspw_implement_amd
BATCHSIZE=8; WORKERS=12; START_GPU=0; GPU_NUM=${GPU_NUM:-2}; TRAINFPS=2; LR=0.002; CROPSIZE=479; EPOCH=120; VAL=False
USE_CLIPDATASET=True
CFG="$(python -c "import $PKG, os; print(os.path.dirname($PKG.__file__))")/config/$CFGNAME"
PREDIR=${PREDIR:-"./imgnetpre/resnet101-imagenet.pth"}
NAME="job_lr${LR}batchsize${BATCHSIZE}_EPOCH${EPOCH}_FPS${TRAINFPS}_arch${ARCH}new124_gpu${GPU_NUM}_480pUSE_CLIPDATASET${USE_CLIPDATASET}"
SAVEROOT=$SAVE/$NAME
echo 'train...'
python -m torch.distributed.run --nnodes=1 --nproc-per-node $GPU_NUM --master-addr 127.0.0.1 -m $PKG.train \
  --cfg $CFG --predir $PREDIR --batchsize $BATCHSIZE --workers $WORKERS --start_gpu $START_GPU --gpu_num $GPU_NUM \
  --dataroot $DATAROOT --trainfps $TRAINFPS --lr $LR --multi_scale True --saveroot $SAVEROOT --totalepoch $EPOCH \
  --cropsize $CROPSIZE --validation $VAL --use_clipdataset $USE_CLIPDATASET
for SPLIT in val test; do
  echo "$SPLIT..."
  python -m $PKG.test --cfg $CFG --start_gpu $START_GPU --dataroot $DATAROOT --saveroot ./saveimg/${NAME}_train \
    --load_en $SAVEROOT/encoder_epoch_$EPOCH.pth --load_de $SAVEROOT/decoder_epoch_$EPOCH.pth --batchsize 2 \
    --is_save True --split $SPLIT
done
