"""TEST INFRASTRUCTURE: runs bench.main() -- unmodified -- with one rank per PROCESS on the CPU, so that the multi-rank control flow
of bench.py (rendezvous, weight-checksum broadcast, barriers, max-over-ranks timing, all-gather of the edited latents, the
frame-sharded mode and its probe, the single JSON line on rank 0, teardown) is exercised without GPUs:

  * `bench.torch` becomes a proxy whose `device(...)` is the CPU and whose `cuda` namespace is a stand-in (wall-clock events);
  * `torch.distributed.init_process_group("nccl")` is answered with gloo;
  * the kernel library is the CPU emulation build and the model is the tiny16 UNet (bench.build_pipeline is replaced here).
Nothing of this exists in bench.py; launched by tests/test_bench_host.py under torch.distributed.run."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from fatezero_amd import _native, build  # noqa: E402


class _Event:
    def __init__(self, enable_timing=False):
        self.t = None

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _Cuda:
    Event = _Event

    @staticmethod
    def is_available():
        return True

    @staticmethod
    def set_device(i):
        pass

    @staticmethod
    def synchronize(*a):
        pass


class _TorchProxy:
    def __init__(self, real):
        self._real = real
        self.cuda = _Cuda()

    def device(self, *a, **k):
        return self._real.device("cpu")

    def __getattr__(self, name):
        return getattr(self._real, name)


def _tiny_pipeline(device, seed=0, model_config=None):
    from fatezero_amd.synthetic import HashTextEncoder, WordTokenizer, init_like_tuned_checkpoint
    from fatezero_amd.video_diffusion.models import UNetPseudo3DConditionModel
    from fatezero_amd.video_diffusion.pipelines.p2p_ddim_spatial_temporal import P2pDDIMSpatioTemporalPipeline
    from fatezero_amd.video_diffusion.schedulers import DDIMScheduler
    torch.manual_seed(seed)
    unet = UNetPseudo3DConditionModel(sample_size=8, in_channels=4, out_channels=4, block_out_channels=(32, 64, 128, 128),
                                      layers_per_block=2, cross_attention_dim=64, attention_head_dim=2, norm_num_groups=8,
                                      **(model_config or {"lora": 16}))
    init_like_tuned_checkpoint(unet, seed)
    pipe = P2pDDIMSpatioTemporalPipeline(vae=None, text_encoder=HashTextEncoder(64), tokenizer=WordTokenizer(), unet=unet.half().eval(),
                                         scheduler=DDIMScheduler())
    pipe.set_progress_bar_config(disable=True)
    return pipe


def main():
    _native.use_test_backend(build.build_emu())
    os.environ.setdefault("FZ_EMU_THREADS", "2")
    bench.torch = _TorchProxy(torch)
    bench.build_pipeline = _tiny_pipeline
    # the calibration launch is the judged 8 x 4096 x 8192 flash shape (minutes on the emulator): a stand-in of the same form
    bench.measure_box = lambda K, device, n=12, hot=False: {"flash_calib_hot_us": 910.0} if hot else {"flash_calib_us": 910.0, "flash_calib_ref_us": bench.FLASH_CALIB_REF_US, "copy_1GiB_TBps": 1.0}
    # 8x8 latents: the blend-word maps of spatial_blend.py:78 only line up at the 512^2 / 576^2 list layouts -> no blend words here
    bench.EDIT_KW = {k: v for k, v in bench.EDIT_KW.items() if k not in ("blend_words", "blend_self_attention", "blend_th")}
    bench.EDIT0_KW = {k: v for k, v in bench.EDIT0_KW.items() if k not in ("blend_words", "blend_self_attention", "blend_th")}
    import torch.distributed as dist
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, **kw: real_init("gloo", **kw)
    bench.main()


if __name__ == "__main__":
    main()
