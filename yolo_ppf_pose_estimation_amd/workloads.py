"""The BASELINE.json configurations as data: which model, which synthetic crops, which matching parameters.

One definition shared by bench.py, the parity tests and the fixture generator
(tests/golden/make_config_fixtures.py), so the three cannot drift apart.  Sizes follow SURVEY.md §8d.

  C2  bottle sampled at 0.036 (2,000 points) vs one 50,000-point crop (seed 12345), presampled, stride 20
  C3  as C2, one crop per rank, seeds 1000 + rank
  C4  bottle sampled at 0.0135 (~10k points) vs one 200,000-point scene (seed 4, two instances)
  C5  four resident models (bottle + box, cylinder, torus at 0.036) x 8 crops per rank (seeds 5000 + 8*rank + c,
      crop c shows model c mod 4), every crop matched against every model
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

from . import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REL_DISTANCE = 0.05
SCENE_STEP = 1.0 / 20.0   # reference stride 20
TOP_K = 5                 # poses kept per crop (CloudProcessing.h:455,508)

C2 = {"model_step": 0.036, "n_points": 50000, "seed": 12345, "n_instances": 1}
C4 = {"model_step": 0.0135, "n_points": 200000, "seed": 4, "n_instances": 2}
C5_MODEL_STEP = 0.036
C5_CROPS_PER_RANK = 8


def bottle() -> np.ndarray:
    """The reference's model (data/bottle_remesh_meter_normalized.ply) as the committed N x 6 array."""
    return np.load(os.path.join(ROOT, "tests", "golden", "bottle_model_xyzn.npy"))


def c2_scene(n_points: int = C2["n_points"]):
    return synth.make_scene(bottle(), n_points=n_points, seed=C2["seed"])[0]


def c3_scene(rank: int, n_points: int = C2["n_points"]):
    return synth.make_scene(bottle(), n_points=n_points, seed=1000 + rank)[0]


def c4_scene(n_points: int = C4["n_points"]):
    return synth.make_scene(bottle(), n_points=n_points, seed=C4["seed"], n_instances=C4["n_instances"])[0]


def c5_models():
    return [bottle(), synth.make_solid("box", 20000, seed=1), synth.make_solid("cylinder", 20000, seed=2),
            synth.make_solid("torus", 20000, seed=3)]


def c5_crops(rank: int = 0, n_points: int = C2["n_points"], n_crops: int = C5_CROPS_PER_RANK, models=None):
    models = models if models is not None else c5_models()
    return [synth.make_scene(models[c % 4], n_points=n_points, seed=5000 + C5_CROPS_PER_RANK * rank + c)[0]
            for c in range(n_crops)]


def cloud_digest(a: np.ndarray) -> str:
    """sha256 of a cloud's float32 bytes: fixtures carry it so a test can tell whether the crop it regenerated is
    bit-identical to the one the oracle saw when the fixture was made."""
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()
