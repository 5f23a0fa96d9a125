"""`python CLIP/frame_acc_tem_con.py` as in the reference (frame accuracy / temporal consistency of ./baselines_results/ours against
CLIP/bench_clean_prompt.yaml): the arithmetic and the CLIP ViT-B/32 encoder are fatezero_amd/metrics.py and fatezero_amd/clip.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fatezero_amd.metrics import main  # noqa: E402

if __name__ == "__main__":
    main()
