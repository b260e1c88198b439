#!/usr/bin/env python3
"""Generates the small golden fixtures committed next to this script.

The reference ships no golden vectors, and it is CUDA C++ that
cannot be imported or built here, so these fixtures pin what CAN be pinned
independently of this repository's own code:
  * fp16_patterns.json — fp64 -> fp16 and fp32 -> fp16 round-to-nearest-even bit
    patterns computed by NUMPY (an implementation independent of float16.hpp
    and of the GPU's conversion instructions), incl. ties, subnormals, overflow.
  * generator.json — the known answers derived in SURVEY.md §8c C-4 from
    libstdc++'s documented `minstd_rand0` / `uniform_real_distribution`
    semantics (values copied from the survey, not produced by this repo's code).
Run: python tests/golden/make_golden.py
"""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def f64_hex(v):
    return struct.pack(">d", float(v)).hex()


def main():
    rng = np.random.default_rng(2024)
    values = np.concatenate([
        rng.uniform(-1, 1, 400),
        rng.uniform(-70000, 70000, 60),
        rng.uniform(-1e-4, 1e-4, 60),
        rng.uniform(-2e-7, 2e-7, 60),
        2.0 ** np.arange(-27, 17, dtype=np.float64),
        # ties and near-ties around representable halves
        np.array([1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -40,
                  1.0 + 2.0 ** -11 - 2.0 ** -40, 65504.0, 65519.999, 65520.0, 65536.0,
                  2.0 ** -25, 2.0 ** -25 * (1 + 2.0 ** -30), 3 * 2.0 ** -25,
                  2.0 ** -14 - 2.0 ** -25, 0.0, -0.0, np.inf, -np.inf]),
    ])
    with np.errstate(over="ignore"):
        half = values.astype(np.float16).view(np.uint16)
        single = values.astype(np.float32)
        half_from_single = single.astype(np.float16).view(np.uint16)
    fixture = {
        "note": "expected = numpy astype(float16) bit patterns (RNE, single rounding)",
        "f64_inputs_hex": [f64_hex(v) for v in values],
        "f64_to_f16_bits": [int(b) for b in half],
        "f32_inputs_bits": [int(b) for b in single.view(np.uint32)],
        "f32_to_f16_bits": [int(b) for b in half_from_single],
    }
    with open(os.path.join(HERE, "fp16_patterns.json"), "w") as f:
        json.dump(fixture, f)

    generator = {
        "note": "SURVEY.md §8c C-4: std::default_random_engine(42) (minstd_rand0), "
                "uniform_real_distribution<double>(-1,1), 2 engine draws per value",
        "first_values": [0.049174203583507659, -0.47338891843144348, -0.60742834883854191,
                         0.024636217393456228, -0.4857967411683386, 0.63097525363607554],
        "first_values_as_float": [0.0491742045, -0.47338891, -0.607428372],
        "gemv_max_size": 24500,
        "gemv_x0": -0.96859211761490849,
        "gemv_res_init0": -0.71172935118387348,
        "dot_n": 1000000,
        "dot_x_last": -0.27059085363363,
        "dot_y_first": -0.73018659478092962,
        "dot_y_last": 0.31734777107476853,
        "dot_fp64": 154.655753144421740,
        "dot_fp64_naive_serial": 154.65575314443419,
        "dot_of_fp32_rounded_copies": 154.6557637953177,
    }
    with open(os.path.join(HERE, "generator.json"), "w") as f:
        json.dump(generator, f, indent=1)


if __name__ == "__main__":
    main()
