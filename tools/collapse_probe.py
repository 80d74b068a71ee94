"""CPU model of the collapsed operator: [up2x] -> sinc -> [delay] -> [down2x] as
ONE time-varying FIR from source frames to target frames, coefficients in
float64.  Checks the index conventions (and the accuracy to expect) against
the oracle before they go into the device code.  Push direction, one call
from the stream's start.  Usage: python tools/collapse_probe.py [quality]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import support  # noqa: E402

TABLES = np.fromfile(os.path.join(ROOT, "data", "drb_tables.bin"), np.float32)
HALF_BAND = {16: 0, 32: 8, 48: 24, 64: 48, 80: 80, 96: 120}
SINC = {16: (168, 64, 16), 32: (1192, 64, 32), 116: (3240, 32, 16), 132: (3752, 32, 32)}


def sinc_position(phase, sd, pull):
    q, r = divmod(phase, sd)
    if pull:
        return q, r / sd
    if r == 0:
        q, r = q - 1, sd
    return q, (r - 1) / sd


def blended(arg, blend):
    offset, rows, taps = SINC[arg]
    bank = TABLES[offset:offset + rows * taps].reshape(rows, taps).astype(np.float64)
    if arg >= 100:
        scaled = blend * 29.0
        row = int(scaled)
        b = float(np.float32(scaled - row))
        w = [-0.5 * b + b * b - 0.5 * b ** 3, 1 - 2.5 * b * b + 1.5 * b ** 3,
             0.5 * b + 2 * b * b - 1.5 * b ** 3, -0.5 * b * b + 0.5 * b ** 3]
        return sum(w[j] * bank[row + j] for j in range(4))
    scaled = blend * 63.0
    row = int(scaled)
    b = float(np.float32(scaled - row))
    return (1 - b) * bank[row] + b * bank[row + 1]


def collapsed_row(f, phase0, sd, td, up, sinc_arg, delay_front, delay_behind, down, pull=False):
    """{source frame: coefficient} of target frame f."""
    taps = sinc_arg % 100
    terms = []
    if down:
        n_d = down // 2
        h_d = TABLES[HALF_BAND[down]:HALF_BAND[down] + n_d].astype(np.float64)
        terms.append((0.5, 2 * f - n_d))
        terms += [(h_d[j], 2 * (f - n_d + j) + 1) for j in range(n_d)]
    else:
        terms.append((1.0, f))
    row = {}
    if up:
        n_u = up // 2
        h_u = TABLES[HALF_BAND[up]:HALF_BAND[up] + n_u].astype(np.float64)
    for weight, i in terms:
        n = i - delay_behind
        q, blend = sinc_position(phase0 + n * td, sd, pull)
        a = blended(sinc_arg, blend)
        for t in range(taps):
            mu = q - taps + t - delay_front
            if not up:
                row[mu] = row.get(mu, 0.0) + weight * a[t]
            elif mu % 2 == 0:
                k = mu // 2 - n_u // 2
                row[k] = row.get(k, 0.0) + weight * a[t]
            else:
                g = (mu - 1) // 2
                for j in range(n_u):
                    k = g - n_u + 1 + j
                    row[k] = row.get(k, 0.0) + weight * a[t] * 2.0 * h_u[j]
    return row


def main():
    quality = int(sys.argv[1]) if len(sys.argv) > 1 else support.GOOD
    src, dst = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (44100, 48000)
    status, plan = support.oracle_plan(src, dst, 1, 256, 16, support.PUSH, quality)
    stages = plan.stages()
    print(stages)
    up = next((arg for kind, arg in stages if kind == "up2x"), 0)
    down = next((arg for kind, arg in stages if kind == "down2x"), 0)
    index = [kind for kind, _ in stages]
    s = index.index("sinc")
    sinc_arg = stages[s][1]
    block = next((arg for kind, arg in stages if kind == "slicer"), 0)
    delay = block // 2
    delay_front = delay if "slicer" in index and index.index("slicer") < s else 0
    delay_behind = delay - delay_front
    in_rate, out_rate = plan.stage_in_rate[s], plan.stage_out_rate[s]
    sd, td = int(2.0 ** 48 / in_rate), int(2.0 ** 48 / out_rate)
    frames = 20000
    x = support.white_noise(1, frames, 5)
    want = support.oracle_push(plan, x)[0]
    xs = x[0].astype(np.float64)
    worst, widths, los = 0.0, [], []
    for f in list(range(0, 400)) + list(range(want.size - 300, want.size)):
        row = collapsed_row(f, sd, sd, td, up, sinc_arg, delay_front, delay_behind, down)
        keys = [k for k, v in row.items() if v != 0.0]
        widths.append(max(keys) - min(keys) + 1)
        los.append(min(keys))
        assert max(keys) < frames
        y = sum(v * xs[k] for k, v in row.items() if k >= 0)
        worst = max(worst, abs(y - float(want[f])))
    print("worst |collapsed f64 - oracle|: %.3g over %d rows; support %d..%d frames" %
          (worst, len(widths), min(widths), max(widths)))
    print("first source frame of rows 0..4:", los[:5], "| of the last rows:", los[-3:])


if __name__ == "__main__":
    main()
