"""Golden VECTORS from the reference's stored figure of its double-integrator closed loop: examples/figures/double_integrator.pdf,
written by examples/1.double_integrator_sim.py:163-203 (N = 2, 12 steps from x0 = (-5, -2), noise = a random vertex of W per step).

The figure is vector graphics: its content stream holds, with six decimals of a PDF point (1.5e-8 state units),

  * the 13 states x_0 .. x_12 of the TZ-DDPC run (the solid black line, :191),
  * the 13 grey polygons Ze_full[t] = Ze[1] of the solve at step t-1, moved by xbar_t, reduced to order <= 3 (:165-168, :82-83),
  * the ZPC comparator's states (dotted red line; another algorithm, kept for completeness).

PDF points -> state units from the axis ticks: the tick marks sit 66.528 pt apart for 1 unit on both axes (equal aspect, set by the
imshow of :178), x = 0 at 627.933669 pt (the last x tick: its label is typeset in the font without the minus sign), y = 0 at
223.48685 pt; x_0 maps to (-5, -2) = the centre of X0 (:49) to the last digit.

What these numbers determine is worked out in tests/refdi.py (inputs and disturbances of every step, the run's identified model
centre, its gain and the magnitudes of its tube, the nominal states and errors).

DATA only: nothing of the reference's source is read or stored.  Run in the build container (the reference is not on the GPU box):

    python tests/golden/extract_reference_di_figure.py        ->  tests/golden/di_reference_figure.npz
"""
import os
import re
import zlib

import numpy as np

SRC = "/root/reference/examples/figures/double_integrator.pdf"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "di_reference_figure.npz")
PT_PER_UNIT = 66.528            # distance of two neighbouring tick marks (1 unit) on either axis
X_ZERO_PT = 627.933669          # x tick labelled 0
Y_ZERO_PT = 223.48685           # y tick labelled 0


def to_units(pts):
    p = np.asarray(pts, float)
    return np.stack([(p[..., 0] - X_ZERO_PT) / PT_PER_UNIT, (p[..., 1] - Y_ZERO_PT) / PT_PER_UNIT], -1)


def main():
    raw = open(SRC, "rb").read()
    streams = re.findall(rb"stream\r?\n(.*?)endstream", raw, re.S)
    text = zlib.decompress(streams[0].rstrip(b"\r\n")).decode("latin1")           # the page's content stream
    ticks_x = sorted({float(a) for a, b in re.findall(r"(-?[\d.]+) 57\.16685 m\n(-?[\d.]+) 53\.66685 l", text)})
    ticks_y = sorted({float(b) for b in re.findall(r"62\.445669 (-?[\d.]+) m\n58\.945669 ", text)})
    assert len(ticks_x) == 9 and len(ticks_y) == 5, (ticks_x, ticks_y)
    assert np.allclose(np.diff(ticks_x), PT_PER_UNIT, atol=1e-6) and np.allclose(np.diff(ticks_y), PT_PER_UNIT, atol=1e-6)
    assert abs(ticks_x[-1] - X_ZERO_PT) < 1e-9 and abs(ticks_y[2] - Y_ZERO_PT) < 1e-9
    grey, salmon, lines = [], [], {}
    for block in re.split(r"\nQ q ", text):
        pts = [(float(a), float(b)) for a, b, _ in re.findall(r"(-?[\d.]+) (-?[\d.]+) ([ml])\n", block + "\n")]
        if not pts:
            continue
        filled = "\nB" in block
        if filled and ("0.8274509804\ng" in block[:160] or "0.8274509804 g" in block[:200]):
            grey.append(to_units(pts))                                               # facecolor 'lightgray' (:167): TZ-DDPC
        elif filled and "0.6274509804" in block[:200]:
            salmon.append(to_units(pts))                                             # 'lightsalmon' (:171): ZPC
        elif len(pts) == 13 and not filled:
            lines["zpc" if "1 0 0 RG" in block[:200] else "tzddpc"] = to_units(pts)
    assert len(grey) == 13 and len(salmon) == 13 and set(lines) == {"zpc", "tzddpc"}
    assert all(len(g) == 8 for g in grey[1:]) and len(grey[0]) == 2                  # Ze_full[0] is the point x0
    polys = np.stack([np.repeat(grey[0][:1], 8, axis=0)] + grey[1:])
    x = lines["tzddpc"]
    assert np.abs(x[0] - np.array([-5.0, -2.0])).max() < 1e-8 and np.abs(polys[0] - x[0]).max() < 1e-8
    np.savez(OUT, x=x, polygons=polys, x_zpc=lines["zpc"], polygons_zpc=np.stack([np.resize(s, (8, 2)) for s in salmon]),
             provenance=np.array("reference artefact examples/figures/double_integrator.pdf (content stream, tick-calibrated); "
                                 "written by tests/golden/extract_reference_di_figure.py; data only"))
    print("wrote", OUT, "states", x.shape, "polygons", polys.shape)


if __name__ == "__main__":
    main()
