"""Make a checkpoint written by this build readable by the REFERENCE's own tools (quick_demo.py, test_gan.py,
demo_inversion.py, demo_interpolation.py): they index `ckpt["cfg"]` by attribute, this build stores it as plain
dicts / lists.  Needs omegaconf (the reference's dependency); run it in the reference's environment:

    python scripts/ckpt_to_upstream.py logs/.../checkpoint_0025000000.pth dustyv2_mi355x.pth
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch

if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    # plain containers + tensors only: weights_only loading suffices, the package itself is not needed
    ckpt = torch.load(src, map_location="cpu", weights_only=True)
    from omegaconf import OmegaConf
    ckpt["cfg"] = OmegaConf.create(ckpt["cfg"])
    torch.save(ckpt, dst)
    print(f"{dst}: cfg re-wrapped as {type(ckpt['cfg']).__name__}; keys {sorted(ckpt)}")
