#!/usr/bin/env python3
"""Numerical model of the bf16 x 6 and f16 x 3 schemes of conv6_kernels.hip (numpy, CPU).

Every fp32 operand is split exactly into three bf16 pieces; of the nine piece products the six of order >= 2^-16 are
accumulated in fp32.  Prints the error of that scheme, of the cheaper three-product scheme and of an ordinary fp32
matmul against a float64 reference, on a GEMM shaped like one output tile of a 64 -> 64 3x3 layer (K = 576)."""
import numpy as np


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    a0 = bf16(x); r = x - a0
    a1 = bf16(r); r2 = r - a1
    a2 = bf16(r2)
    return a0, a1, a2


def mm32(a, b):
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)


def six_product(A, B):
    a, b = split3(A), split3(B)
    # smallest terms first, as the kernel issues them
    return (mm32(a[2], b[0]) + mm32(a[0], b[2]) + mm32(a[1], b[1]) + mm32(a[1], b[0]) + mm32(a[0], b[1]) + mm32(a[0], b[0]))


def three_product(A, B):
    a, b = split3(A), split3(B)
    return mm32(a[1], b[0]) + mm32(a[0], b[1]) + mm32(a[0], b[0])


def split_f16(x, scale=2048.0):
    """x = hi + lo / scale with hi = f16(x), lo = f16((x - hi) * scale)  (k_conv6<.., 2>: split_pair_h); returned as float32"""
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(scale)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def f16x3_product(A, B, scale=2048.0):
    """hi*hi in one fp32 accumulator, hi*lo + lo*hi in a second one that is scaled back at the end; lo*lo (2^-22) dropped"""
    ah, al = split_f16(A, scale)
    bh, bl = split_f16(B, scale)
    return mm32(ah, bh) + (mm32(ah, bl) + mm32(al, bh)) * np.float32(1.0 / scale)


def f16x3_representation_error(A, B, scale=2048.0):
    """error of the three kept products alone (accumulated in float64), against the exact product sum"""
    d = lambda x: x.astype(np.float64)
    ah, al = split_f16(A, scale)
    bh, bl = split_f16(B, scale)
    x = d(ah) @ d(bh) + (d(ah) @ d(bl) + d(al) @ d(bh)) / scale
    return rel(x, d(A) @ d(B))


def rel(x, ref):
    return float(np.linalg.norm(x.astype(np.float64) - ref) / np.linalg.norm(ref))


def main():
    rng = np.random.default_rng(0)
    M, K, N = 64, 576, 256
    A = (rng.standard_normal((M, K)) * 0.05).astype(np.float32)
    B = np.maximum(rng.standard_normal((K, N)), 0).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    a = split3(A)
    print("split exact      :", bool(np.all((a[0].astype(np.float64) + a[1] + a[2]) == A.astype(np.float64))))
    print("fp32 matmul      : %.2e" % rel(mm32(A, B), ref))
    print("bf16 x 6 products: %.2e" % rel(six_product(A, B), ref))
    print("bf16 x 3 products: %.2e" % rel(three_product(A, B), ref))
    print("f16 x 3 products : %.2e  (the kept products alone, exact accumulation: %.2e)" % (rel(f16x3_product(A, B), ref), f16x3_representation_error(A, B)))


if __name__ == "__main__":
    main()
