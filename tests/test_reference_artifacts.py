"""CPU: the checkpoint surface against the reference's own shipped artefacts (resources/DDPG/<task>/
{actor,critic}.pth, obs.yaml, dg.yaml) where the reference tree is present — this container only; on the
GPU box the tree does not exist and the test is skipped.  Nothing is copied: files are read in place."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import ROOT  # noqa: F401

REF = "/root/reference/resources/DDPG"
TASKS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(REF, "*")) if os.path.isdir(p))

pytestmark = pytest.mark.skipif(not TASKS, reason="reference tree not present")


@pytest.mark.parametrize("task", TASKS or ["none"])
def test_state_dict_layout_matches_reference_checkpoints(gcrl, task):
    from gcrl_amd.src.model import Actor, Critic
    for fname, cls in (("actor.pth", Actor), ("critic.pth", Critic)):
        path = os.path.join(REF, task, fname)
        if not os.path.exists(path):
            continue
        sd = torch.load(path, map_location="cpu", weights_only=True)
        keys = list(sd.keys())
        first, last = sd[keys[0]], sd[keys[-2]]
        layers = len(keys) // 2 - 1                     # hidden Linear layers
        view = cls(None, fname, int(first.shape[1]), int(first.shape[0]), int(last.shape[0]), layers)
        layout = view._param_layout()
        assert [k for k, _ in layout] == keys                                   # names and order of module.parameters()
        assert [tuple(s) for _, s in layout] == [tuple(v.shape) for v in sd.values()]   # [out, in] weights
        flat = view.join(sd)                                                     # -> the engine's flat vector
        back = view.split(flat)
        for k in keys:
            assert np.array_equal(back[k], sd[k].numpy())


@pytest.mark.parametrize("task", TASKS or ["none"])
def test_normaliser_yaml_loads(gcrl, task):
    from gcrl_amd.src.utils import RunningNormalizer
    for fname in ("obs.yaml", "dg.yaml"):
        path = os.path.join(REF, task, fname)
        if not os.path.exists(path):
            continue
        import yaml
        raw = yaml.safe_load(open(path))
        n = RunningNormalizer(len(raw["mean"]))
        n.load(path)
        assert np.allclose(n.mean, np.asarray(raw["mean"], dtype=np.float64))
        x = np.zeros((2, len(raw["mean"])))
        assert np.all(np.abs(n.normalize(x)) <= raw["clip_range"] + 1e-9)
