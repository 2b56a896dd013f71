"""cvpr2021_vspw_implement_amd — MI355X-native (gfx950) implementation of the temporal-context video-segmentation
hot path of sssdddwww2/CVPR2021_VSPW_Implement: ResNet-dilated + PPM/OCR encoder-decoders and the clip-level
Temporal Context Blending heads, executed by hand-written HIP kernels behind the reference's
ModelBuilder / SegmentationModule / Clip_PSP / ClipOCRNet surface.  See DESIGN.md."""
__version__ = "0.1.0"
