import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_genome():
    z = np.load(os.path.join(GOLDEN, "genome.npz"))
    return z["packed"], int(z["ref_size"])


def cases(kind):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        name = os.path.basename(f)[:-4]
        if name == "genome":
            continue
        prefix = {"distance": "dist_", "levdiag": "lev_", "align": "align_", "indel": "indel_",
                  "clip": "clip_", "cf": "cf_", "samcigar": "samcigar_", "clipillegal": "clipillegal_",
                  "clippairs": "clippairs_", "softclip": "softclip_", "finish": "finish_", "groups": "groups_", "saloop": "saloop_", "rescue": "rescue_"}[kind]
        if name.startswith(prefix):
            out.append(name)
    return out


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def cigar_bytes(cig, i):
    return bytes(cig[i]).split(b"\0")[0]


def scoring(z, module):
    """The fixture's scoring_t as module.Scoring (module = oracle.bindings or whisper_amd.api), None = the defaults."""
    if "scoring" not in z.files:
        return None
    return module.Scoring(*[float(v) for v in z["scoring"]])
