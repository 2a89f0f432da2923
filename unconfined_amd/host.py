"""Host-side pieces of the driver that are plain bookkeeping (no hot-path compute).

These mirror what ``program Driver`` does around the (t,r) loop body
(reference driver.f90:234-273): the observation-screen average (quirk Q2), the
scaling back to dimensional heads and the output rows.
"""
from __future__ import annotations

import numpy as np


def screen_average_np(h: np.ndarray, dk) -> np.ndarray:
    """driver.f90:234-243.  ``h`` is [npts, nz]; returns [npts].

    Not a textbook trapezoid:
    ``(h(1) + 2*sum(h(2:zOrd)) + h(zOrd)) / (2*zOrd)`` -- reproduced literally,
    including the left-to-right order of the additions.
    """
    h = np.asarray(h, dtype=np.float64)
    if dk.timeseries and (not dk.piezometer) and dk.zOrd > 1:
        zo = dk.zOrd
        s = h[:, 1].copy()
        for j in range(2, zo):
            s = s + h[:, j]
        return ((h[:, 0] + 2.0 * s) + h[:, zo - 1]) / (2 * zo)
    return h[:, 0].copy()
