"""Robot descriptions -- stand-in for the reference's per-robot ``config-<robot>-<instance>.py``
files (scripts/vfclik:80-81), which are not in the reference tree.

The numbers below are this build's own: the public KUKA LWR 4+ DH table and joint ranges (the
reference's default robot name is ``lwr``, scripts/vfclik:42, src/command_mixer.py:96).  They are
not claimed to equal whatever arcospyu's config holds.
"""
import math

import numpy as np

from .chain import Chain

_D2R = math.pi / 180.0

# (a, alpha, d, theta_offset), standard DH
_LWR_DH = [
    (0.0, math.pi / 2, 0.310, 0.0),
    (0.0, -math.pi / 2, 0.0, 0.0),
    (0.0, -math.pi / 2, 0.400, 0.0),
    (0.0, math.pi / 2, 0.0, 0.0),
    (0.0, math.pi / 2, 0.390, 0.0),
    (0.0, -math.pi / 2, 0.0, 0.0),
    (0.0, 0.0, 0.078, 0.0),
]
_LWR_LIM = np.array([170, 120, 170, 120, 170, 120, 170], dtype=float) * _D2R


def lwr():
    """7-DOF KUKA LWR 4+ (configs C1-C4 of BASELINE.json)."""
    return Chain.from_dh(_LWR_DH, -_LWR_LIM, _LWR_LIM, name="lwr")


def lwr_dual14():
    """14-DOF chain of BASELINE config C5: two LWR chains in series, one flange."""
    return lwr().concat(lwr(), name="lwr_dual14")


def powercube6():
    """A 6-DOF arm (the reference also drives a 6-joint 'powercube', old/system_start.sh.old:244-247).
    Geometry is a generic 6R elbow arm of this build's choosing."""
    dh = [
        (0.0, math.pi / 2, 0.30, 0.0),
        (0.35, 0.0, 0.0, 0.0),
        (0.0, math.pi / 2, 0.0, 0.0),
        (0.0, -math.pi / 2, 0.30, 0.0),
        (0.0, math.pi / 2, 0.0, 0.0),
        (0.0, 0.0, 0.10, 0.0),
    ]
    lim = np.array([170, 120, 150, 170, 120, 170], dtype=float) * _D2R
    return Chain.from_dh(dh, -lim, lim, name="powercube6")


def by_name(name):
    return {"lwr": lwr, "lwr_dual14": lwr_dual14, "powercube6": powercube6}[name]()
