"""VSPW datasets of the reference's dataset2.py, split at the point where pixels start to be touched.

The reference's `__getitem__` decodes T frames with PIL and then flips / rescales / pads / crops / normalises them on
the CPU (dataset2.py:986-1048, :783-850, :247-342, :442-490).  Here `__getitem__` does the file listing, the random
draws (same numpy / `random` call sequence, so the same seeds pick the same frames, flip, scale and crop window) and the
JPEG/PNG decode - and returns the decoded uint8 frames plus a `FrameSpec`; `DeviceTransform` then runs the whole pixel
pipeline on the GPU (csrc/data.hip, bit-exact with the reference's PIL + numpy + torchvision chain) and hands the model
the same feed tensors, already NHWC and resident in HBM.  There is no CPU implementation of the pixel pipeline here.

On-disk format (VSPW): `<root>/{train,val,test}.txt` one video name per line; `<root>/data/<video>/origin/*.jpg`,
`<root>/data/<video>/mask/<frame>.png` (`mask_42label` with --lesslabel).
"""
import ctypes
import os
import random

import numpy as np
import torch
from PIL import Image

from . import _C

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def dilation_list(list_, num):
    """dataset2.py:23-30"""
    newlist = []
    a = np.random.choice(list(range(num)))
    for k in range(len(list_)):
        if k % (num + 1) == a:
            newlist.append(list_[k])
    return newlist


def dilation_lists(list_, num):
    """dataset2.py:143-151"""
    return [[list_[k] for k in range(len(list_)) if k % (num + 1) == a] for a in range(num + 1)]


class FrameSpec(object):
    """What the reference would have done to every frame of this sample, decided on the host, applied on the device."""

    __slots__ = ("flip", "new_h", "new_w", "pad_h", "pad_w", "crop_y", "crop_x", "out_h", "out_w")

    def __init__(self, h, w, flip=0, new_hw=None, crop=None):
        self.flip = int(flip)
        self.new_h, self.new_w = (h, w) if new_hw is None else new_hw
        if crop is None:  # whole (resized) frame
            self.pad_h = self.pad_w = self.crop_y = self.crop_x = 0
            self.out_h, self.out_w = self.new_h, self.new_w
        else:
            self.pad_h, self.pad_w, self.crop_y, self.crop_x, self.out_h, self.out_w = crop


class RawSample(object):
    """Decoded frames of one sample: frames[t] uint8 [h,w,3], masks[t] uint8 [h,w], one FrameSpec, frame names."""

    __slots__ = ("frames", "masks", "spec", "names")

    def __init__(self, frames, masks, spec, names):
        self.frames, self.masks, self.spec, self.names = frames, masks, spec, names


def _open_rgb(path, convert):
    im = Image.open(path)
    if convert:
        im = im.convert("RGB")
    a = np.array(im)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("%s: expected an RGB frame, got array shape %s" % (path, a.shape))
    return a


def _open_mask(path):
    a = np.array(Image.open(path))
    if a.ndim != 2 or a.dtype != np.uint8:
        raise ValueError("%s: expected a single-channel uint8 label map, got %s %s" % (path, a.shape, a.dtype))
    return a


class _VSPWBase(torch.utils.data.Dataset):
    def _mask_dir(self):
        return "mask_42label" if getattr(self.args, "lesslabel", False) else "mask"

    def _load(self, video, imgname, convert=True):
        img = _open_rgb(os.path.join(self.dataroot, "data", video, "origin", imgname), convert)
        seg = _open_mask(os.path.join(self.dataroot, "data", video, self._mask_dir(), imgname.split(".")[0] + ".png"))
        return img, seg


class _TrainBase(_VSPWBase):
    """Shared constructor / augmentation draw of BaseDataset_clip and BaseDataset_longclip (dataset2.py:658-696,
    853-898, 921-947)."""

    def __init__(self, args, split="train"):
        self.cropsize = (args.cropsize, args.cropsize)
        self.dataroot = args.dataroot
        self.trainfps = getattr(args, "trainfps", 1)
        self.clipnum = args.clip_num
        self.split = split
        with open(os.path.join(self.dataroot, self.split + ".txt")) as f:
            self.videolists = [line[:-1] for line in f.readlines()]
        self.args = args
        self.scale = [0.8, 1., 1.5, 2.0]
        self.imgdic = {}
        for video in self.videolists:
            self.imgdic[video] = sorted(os.listdir(os.path.join(self.dataroot, "data", video, "origin")))

    def __len__(self):
        return len(self.videolists)

    def _finish(self, video, names, flip_flag, scale):
        """Decode the chosen frames; fix the resize target and the crop window exactly where the reference's
        `__getitem__` would draw them (rand_crop's two random.randint calls come after all frames are loaded)."""
        frames, masks = [], []
        for n in names:
            img, seg = self._load(video, n)
            frames.append(img)
            masks.append(seg)
        h, w = frames[0].shape[:2]
        train = self.split == "train"
        new_hw = None
        if train and getattr(self.args, "multi_scale", False) and scale != 1.:
            new_hw = (int(h * scale), int(w * scale))
        rh, rw = new_hw if new_hw is not None else (h, w)
        crop = None
        if train:
            padw = self.cropsize[1] - rw if rw < self.cropsize[1] else 0
            padh = self.cropsize[0] - rh if rh < self.cropsize[0] else 0
            ph, pw = rh + 2 * padh, rw + 2 * padw
            x = random.randint(0, pw - self.cropsize[1])
            y = random.randint(0, ph - self.cropsize[0])
            crop = (padh, padw, y, x, self.cropsize[0], self.cropsize[1])
        spec = FrameSpec(h, w, flip_flag if train else 0, new_hw, crop)
        return RawSample(frames, masks, spec, list(names))


class BaseDataset_longclip(_TrainBase):
    """dataset2.py:852-1048: one clip per video - a random start frame and the frames `dilation2` offsets after it."""

    def __init__(self, args, split="train"):
        super().__init__(args, split)
        self.dilation = [int(d) for d in args.dilation2.split(",")]
        assert len(self.dilation) + 1 == self.clipnum

    def __getitem__(self, idx):
        video = self.videolists[idx]
        imglist = self.imgdic[video]
        if np.random.random() < 0.5:
            imglist = imglist[::-1]
        imglist_s = imglist[:-self.dilation[-1]]
        while len(imglist_s) < 1:
            imglist.append(imglist[-1])
            imglist_s = imglist[:-self.dilation[-1]]
        idx = np.random.choice(list(range(len(imglist_s))))
        this_step = [idx] + [idx + dil for dil in self.dilation]
        flip_flag = np.random.choice([0, 1])
        scale = np.random.choice(self.scale)
        return self._finish(video, [imglist[i] for i in this_step], flip_flag, scale)


class BaseDataset_clip(_TrainBase):
    """dataset2.py:657-850: clip_num consecutive frames of one of the (dilation_num+1)-strided sub-sequences."""

    def __init__(self, args, split="train"):
        super().__init__(args, split)
        self.dilation = args.dilation_num

    def __getitem__(self, idx):
        video = self.videolists[idx]
        imglists = dilation_lists(self.imgdic[video], self.dilation)
        for _ in range(10):
            idd = np.random.choice(list(range(len(imglists))))
            imglist = imglists[idd]
            if len(imglist) > self.clipnum:
                break
        if len(imglist) <= self.clipnum:
            for _ in range(self.clipnum + 1 - len(imglist)):
                imglist.append(imglist[-1])
        imgidxs_ = list(range(len(imglist)))[:-self.clipnum]
        imgid = np.random.choice(imgidxs_, 1)[0]
        flip_flag = np.random.choice([0, 1])
        scale = np.random.choice(self.scale)
        return self._finish(video, [imglist[i] for i in range(imgid, imgid + self.clipnum)], flip_flag, scale)


class BaseDataset(_TrainBase):
    """dataset2.py:494-650: the PER-FRAME dataset (train.py's feed for the per-frame PSPNet / OCRNet heads): every
    int(15 / trainfps)-th frame of every video of the split (val: every 15th); train: flip, multi-scale resize, pad to
    the crop size, random crop - one frame, same draw order as the clip datasets; val: the whole frame."""

    def __init__(self, args, split="train"):
        clip_num = getattr(args, "clip_num", 1)
        if not hasattr(args, "clip_num"):
            args.clip_num = clip_num
        super().__init__(args, split)
        if getattr(args, "train_filter", False):
            self.cropsize = (480, 720)
        if self.split == "val":
            self.trainfps = 1
        num = int(15. / self.trainfps)
        self.imglist = [(video, name) for video in self.videolists
                        for k, name in enumerate(self.imgdic[video]) if k % num == 0]

    def __len__(self):
        return len(self.imglist)

    def __getitem__(self, idx):
        video, name = self.imglist[idx]
        flip_flag, scale = 0, 1.
        if self.split == "train":
            flip_flag = np.random.choice([0, 1])
            if getattr(self.args, "multi_scale", False):
                scale = np.random.choice(self.scale)
        return self._finish(video, [name], flip_flag, scale)


class _TestBase(_VSPWBase):
    def __init__(self, dataroot, video, args, is_train=False):
        self.dataroot = dataroot
        self.video = video
        self.clip_num = args.clip_num
        self.args = args
        self.is_train = is_train
        self.imglist = sorted(os.listdir(os.path.join(self.dataroot, "data", video, "origin")))
        self.imglist2 = [self.imglist[k] for k in range(len(self.imglist)) if k % 15 == 0] if is_train else []

    def __len__(self):
        return len(self.imglist2) if self.is_train else len(self.imglist)

    def _sample(self, names, imagenames):
        """names[0] is the target frame, the rest its clip; no augmentation (dataset2.py:262-267,457-462: no
        convert('RGB') on this path)."""
        frames, masks = [], []
        for n in names:
            img, seg = self._load(self.video, n, convert=False)
            frames.append(img)
            masks.append(seg)
        h, w = frames[0].shape[:2]
        return RawSample(frames, masks, FrameSpec(h, w), imagenames)


class TestDataset(_VSPWBase):
    """dataset2.py:34-141: the PER-FRAME test dataset (test.py's feed): every frame of one video, no augmentation;
    `use_720p` resizes frame and mask to 1080 x 720 (PIL bilinear / nearest) first.  names = the mask's file name."""

    def __init__(self, dataroot, video, args):
        self.dataroot = dataroot
        self.video = video
        self.args = args
        self.imglist = sorted(os.listdir(os.path.join(self.dataroot, "data", video, "origin")))

    def __len__(self):
        return len(self.imglist)

    def __getitem__(self, idx):
        name = self.imglist[idx]
        img, seg = self._load(self.video, name)
        h, w = img.shape[:2]
        new_hw = (720, 1080) if getattr(self.args, "use_720p", False) else None
        return RawSample([img], [seg], FrameSpec(h, w, 0, new_hw), name.split(".")[0] + ".png")


class TestDataset_longclip(_TestBase):
    """dataset2.py:344-490: frame `index` plus the frames at +dilation2 offsets (mirrored backwards at the end)."""

    def __init__(self, dataroot, video, args, is_train=False):
        super().__init__(dataroot, video, args, is_train)
        self.dilation = [int(d) for d in args.dilation2.split(",")]
        assert len(self.dilation) + 1 == self.clip_num

    def __getitem__(self, index):
        img = self.imglist[index]
        names = [img]
        for dil in self.dilation:
            idx = index - dil if index + self.dilation[-1] >= len(self.imglist) else index + dil
            names.append(self.imglist[idx])
        return self._sample(names, img)


class TestDataset_clip(_TestBase):
    """dataset2.py:154-342: the target frame and its neighbours inside its dilation sub-sequence."""

    def __init__(self, dataroot, video, args, is_train=False):
        super().__init__(dataroot, video, args, is_train)
        self.dilation = args.dilation_num
        self.dilists = dilation_lists(self.imglist, self.dilation)

    def __getitem__(self, index):
        img = self.imglist2[index] if self.is_train else self.imglist[index]
        nl3d = getattr(self.args, "method", "") == "nonlocal3d"
        for dilist in self.dilists:
            if img in dilist:
                thelist = dilist
        imgindex = thelist.index(img)
        add = int(self.clip_num / 2) if self.clip_num % 2 == 0 else int((self.clip_num - 1) / 2)
        addleft = add
        addright = add - 1 if self.clip_num % 2 == 0 else add
        if imgindex - addleft < 0:
            start = 0
            end = min(start + self.clip_num, len(thelist))
        elif imgindex + addright >= len(thelist):
            end = len(thelist)
            start = max(end - self.clip_num, 0)
        else:
            start = imgindex - addleft
            end = start + self.clip_num
        names = [img]
        imagenames = [] if nl3d else img
        if end - start < 2:
            names.append(img)
        else:
            for idx in range(start, end):
                if not nl3d and idx == imgindex:
                    continue
                names.append(thelist[idx])
                if nl3d:
                    imagenames.append(thelist[idx])
        return self._sample(names, imagenames)


class TwoDataset(torch.utils.data.Dataset):
    """dataset2.py:1052-1246, imported by train.py:16 (`--usetwodata`: VSPW mixed with a second, ADE-style tree; every
    script of the reference runs with USETWODATA=False).  Not part of the VSPW path: importable, raises at construction."""

    def __init__(self, *a, **k):
        raise NotImplementedError("dataset2.TwoDataset (--usetwodata) is outside the VSPW hot-path scope (SURVEY.md §8)")


MultiScaleTrainDataset = TwoDataset


def collate_raw(samples):
    """DataLoader collate_fn: keep the decoded samples as they are (the batch is assembled on the device)."""
    return list(samples)


# ------------------------------------------------------------------------------------------------ Pillow tables
_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_tables(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (libImaging/Resample.c) in the same
    double-precision operation order: (bounds [out,2] int32, coefficients [out,ksize] int32, ksize)."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws = []
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w = 1.0 - a if a < 1.0 else 0.0
            ws.append(w)
            ww += w
        for x in range(xmax):
            k = ws[x] / ww if ww != 0.0 else ws[x]
            kk[xx, x] = int(-0.5 + k * (1 << _PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def pil_nearest_table(in_size, out_size):
    """Pillow's nearest-neighbour resize (ImagingScaleAffine): the source index is the running sum a/2 + a + a + ...
    truncated, accumulated in double exactly as the C loop does."""
    a = float(in_size) / out_size
    tab = np.zeros(out_size, dtype=np.int32)
    xo = 0.0 + a * 0.5
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        tab[x] = min(max(xin, 0), in_size - 1)
        xo += a
    return tab


class DeviceTransform(object):
    """Runs the pixel pipeline of a batch of RawSamples on `device` and returns what the reference's DataLoader would
    have produced (default collate): `imgs` = T tensors [B,3,H,W] (NHWC memory), `labels` = T tensors [B,1,H,W]."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceTransform: the input pipeline runs only on the GPU (got %s); there is no CPU "
                               "fallback" % (self.device,))
        self._mean = (ctypes.c_float * 3)(*MEAN)
        self._std = (ctypes.c_float * 3)(*STD)
        self._tables = {}

    def _dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device, non_blocking=True)

    def _table(self, kind, n_in, n_out):
        key = (kind, n_in, n_out)
        if key not in self._tables:
            if kind == "bilinear":
                b, k, ks = pil_bilinear_tables(n_in, n_out)
                self._tables[key] = (self._dev(b), self._dev(k), ks)
            else:
                self._tables[key] = self._dev(pil_nearest_table(n_in, n_out))
        return self._tables[key]

    def _resize(self, img, seg, spec, st):
        h, w = img.shape[:2]
        nh, nw = spec.new_h, spec.new_w
        if (nh, nw) == (h, w):
            return img, seg
        # the reference flips the PIL image, then resizes it: the first pass reads its source mirrored
        flip = spec.flip
        cur, ch, cw = img, h, w
        if nw != w:  # horizontal pass first (ImagingResample)
            b, k, ks = self._table("bilinear", w, nw)
            tmp = torch.empty((h, nw, 3), dtype=torch.uint8, device=self.device)
            _C.call("vspw_resample_u8", cur.data_ptr(), tmp.data_ptr(), b.data_ptr(), k.data_ptr(), ks, h, w, h, nw, 3,
                    0, flip, st)
            cur, cw, flip = tmp, nw, 0
        if nh != h:
            b, k, ks = self._table("bilinear", h, nh)
            tmp = torch.empty((nh, cw, 3), dtype=torch.uint8, device=self.device)
            _C.call("vspw_resample_u8", cur.data_ptr(), tmp.data_ptr(), b.data_ptr(), k.data_ptr(), ks, ch, cw, nh, cw,
                    3, 1, flip, st)
            cur = tmp
        xt, yt = self._table("nearest", w, nw), self._table("nearest", h, nh)
        seg2 = torch.empty((nh, nw), dtype=torch.uint8, device=self.device)
        _C.call("vspw_gather_u8", seg.data_ptr(), seg2.data_ptr(), xt.data_ptr(), yt.data_ptr(), w, nh, nw, spec.flip, st)
        return cur, seg2

    def __call__(self, samples, frames_as_batch=False):
        """frames_as_batch: ONE tensor pair holding all T x B frames, frame-major - what train.py:41-44 builds with
        torch.cat(clip_imgs, dim=0) when the clip dataset feeds a per-frame model - written in place, no concatenation."""
        B = len(samples)
        T = len(samples[0].frames)
        s0 = samples[0].spec
        oh, ow = s0.out_h, s0.out_w
        for s in samples:
            if (s.spec.out_h, s.spec.out_w) != (oh, ow) or len(s.frames) != T:
                raise ValueError("samples of one batch must produce frames of one size (use a crop, or batch size 1)")
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if frames_as_batch:
            allimg = torch.empty((T * B, oh, ow, 3), dtype=torch.float32, device=self.device)
            alllab = torch.empty((T * B, 1, oh, ow), dtype=torch.float32, device=self.device)
            imgs = [allimg[t * B:(t + 1) * B] for t in range(T)]
            labs = [alllab[t * B:(t + 1) * B] for t in range(T)]
        else:
            imgs = [torch.empty((B, oh, ow, 3), dtype=torch.float32, device=self.device) for _ in range(T)]
            labs = [torch.empty((B, 1, oh, ow), dtype=torch.float32, device=self.device) for _ in range(T)]
        for b, s in enumerate(samples):
            sp = s.spec
            for t in range(T):
                img, seg = self._dev(s.frames[t]), self._dev(s.masks[t])
                resized = (sp.new_h, sp.new_w) != tuple(img.shape[:2])
                img, seg = self._resize(img, seg, sp, st)
                flip = 0 if resized else sp.flip
                _C.call("vspw_frame_transform", img.data_ptr(), seg.data_ptr(), sp.new_h, sp.new_w, flip, sp.pad_h,
                        sp.pad_w, sp.crop_y, sp.crop_x, oh, ow, self._mean, self._std, imgs[t][b].data_ptr(),
                        labs[t][b].data_ptr(), st)
        if frames_as_batch:
            return [allimg.permute(0, 3, 1, 2)], [alllab]
        return [i.permute(0, 3, 1, 2) for i in imgs], labs
