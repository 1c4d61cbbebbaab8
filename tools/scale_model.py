#!/usr/bin/env python3
"""The arithmetic behind DESIGN section 10.4: weak-scaling efficiency of the ray-DP step on N GPUs of one node from MEASURED single-box
quantities and an ASSUMED ring all-reduce bus bandwidth -- nothing here is a measurement of more than one GPU.

  step_1        the plain single-GPU step (bench.py, N = 1)                                          [ms]
  step_dp       the ray-DP recording in a one-rank RCCL world (PVD_DP_FORCE=1 PVD_DP_PIPELINE=2):    [ms]
                what every rank executes besides waiting for the exchange
  bytes         the compact gradient that crosses the links (touched rows, fp32)                     [MB]
  ring all-reduce of B bytes over N ranks: 2 (N - 1) / N * B / busbw + 2 (N - 1) * hop latency
  efficiency    step_1 / (step_dp + t_AR - hidden), hidden = what of the exchange an overlap could cover

  python tools/scale_model.py [--step1 0.272] [--step-dp 0.316] [--mb 13.3] [--hop-us 4]"""
import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--step1", type=float, default=0.272)
ap.add_argument("--step-dp", type=float, default=0.316)
ap.add_argument("--mb", type=float, default=13.3)
ap.add_argument("--hop-us", type=float, default=4.0)
a = ap.parse_args()


def t_ar_us(n, mb, busbw_gbs):
    return 2.0 * (n - 1) / n * mb * 1e6 / (busbw_gbs * 1e9) * 1e6 + 2 * (n - 1) * a.hop_us


def t_direct_us(n, mb, link_gbs=76.8, lat_us=6.0):
    """two-shot all-reduce on a fully connected node: reduce-scatter then all-gather, every GPU sends B / n to each
    of its n - 1 peers over its own link at the same time: 2 x (B / n) / link bandwidth (one direction of one xGMI link) + two latencies"""
    return 2.0 * (mb * 1e6 / n) / (link_gbs * 1e9) * 1e6 + 2 * lat_us


print("step_1 %.3f ms, ray-DP recording %.3f ms, %.1f MB on the wire in fp32, %.0f us per ring hop" % (a.step1, a.step_dp, a.mb, a.hop_us))
print("%-44s %8s %8s %8s" % ("", "N = 2", "N = 4", "N = 8"))
for label, mb, hidden in (("fp32 wire, exchange not overlapped (shipped)", a.mb, 0.0),
                          ("16-bit wire (PVD_DP_WIRE, opt-in)", a.mb / 2, 0.0),
                          ("fp32 wire, two of three buckets under the scatter", a.mb, 66.0),
                          ("16-bit wire + bucket overlap", a.mb / 2, 66.0)):
    for bw in (100.0, 150.0, 200.0):
        eff = []
        for n in (2, 4, 8):
            t = max(t_ar_us(n, mb, bw) - hidden, 0.0)
            eff.append(a.step1 * 1e3 / (a.step_dp * 1e3 + t))
        print("%-44s %8.2f %8.2f %8.2f   (bus bandwidth %3.0f GB/s: t_AR(8) = %3.0f us)" % (label, eff[0], eff[1], eff[2], bw, t_ar_us(8, mb, bw)))

for label, mb, hidden in (("direct two-shot over all 7 links, fp32", a.mb, 0.0), ("direct two-shot, 16-bit wire", a.mb / 2, 0.0),
                          ("direct two-shot, fp32, bucket overlap", a.mb, 66.0)):
    eff = [a.step1 * 1e3 / (a.step_dp * 1e3 + max(t_direct_us(n, mb) - hidden, 0.0)) for n in (2, 4, 8)]
    print("%-44s %8.2f %8.2f %8.2f   (76.8 GB/s per link and direction: t_AR(8) = %3.0f us)" % (label, eff[0], eff[1], eff[2], t_direct_us(8, mb)))
