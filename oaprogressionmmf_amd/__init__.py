"""oaprogressionmmf_amd -- MI355X-native (gfx950) implementation of koafusion's multimodal train-step hot
path behind the reference's own registries (dict_models / dict_fes / dict_losses / dict_optimizers /
dict_schedulers).  Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed only);
all arithmetic runs in hand-written HIP kernels reached through the C ABI in include/koaf.h."""
__version__ = "0.1.1"
