"""align3r_amd: MI355X-native pair-forward + global-alignment engine behind Align3R's Python API."""
import importlib
import sys

__version__ = "0.1.0"

_MIRRORED = ["dust3r", "dust3r.model", "dust3r.inference", "dust3r.image_pairs", "dust3r.cloud_opt",
             "dust3r.cloud_opt.optimizer", "dust3r.cloud_opt.commons", "dust3r.cloud_opt_flow",
             "dust3r.cloud_opt_flow.optimizer", "dust3r.utils", "dust3r.utils.device", "dust3r.utils.image_pose",
             "dust3r.utils.goem_opt", "dust3r.cloud_opt.pair_viewer", "dust3r.cloud_opt.init_im_poses"]


def install_as_dust3r():
    """Make ``import dust3r.inference`` etc. resolve to this package's mirror modules; ``from third_party.raft import load_RAFT``
    (dust3r/cloud_opt_flow/optimizer.py:13) to the HIP flow network's loader."""
    for name in _MIRRORED:
        sys.modules[name] = importlib.import_module("align3r_amd." + name)
    import types
    tp = sys.modules.setdefault("third_party", types.ModuleType("third_party"))
    tp.raft = sys.modules["third_party.raft"] = importlib.import_module("align3r_amd.raft")
