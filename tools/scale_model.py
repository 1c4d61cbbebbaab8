#!/usr/bin/env python3
"""The arithmetic behind DESIGN's multi-GPU section: weak-scaling efficiency of the ray-DP step on N GPUs of one node from MEASURED
single-box quantities and ASSUMED link rates -- nothing here is a measurement of more than one GPU (every gpurun box has one).

  step_1       the plain single-GPU step (bench.py, N = 1)                                                         [ms]
  step_ar      the ray-DP recording, all-reduce form, in a one-rank RCCL world (PVD_DP_FORCE=1 PVD_DP_PIPELINE=2):  [ms]
               what every rank executes besides waiting for the wire (profiles/r06_dp_one_rank_forms.txt)
  step_sh      the same with the sharded update (PVD_DP_EXCHANGE=sharded); at N ranks its AdamW part B walks 1/N of the
               touched rows: step_sh - part_b * (1 - 1/N)
  bytes        the compact gradient that crosses the links (touched rows, fp32)                                    [MB]
  wire, ring   2 (N - 1) / N * B / busbw + 2 (N - 1) hops          (all-reduce; reduce-scatter + all-gather: the same bytes, one more launch)
  wire, direct reduce-scatter then all-gather on a fully connected node, every GPU sending B / N to each peer over its own xGMI link
               at once: 2 * (B / N) / (one direction of one link) + two latencies     (/opt/skills guide: 7 links x ~153 GB/s per GPU)
  efficiency   step_1 / (step_form(N) + wire)          -- the exchange is NOT overlapped with anything (it sits between the table
               scatter and the update: both ends are on the step's critical chain)

  python tools/scale_model.py [--step1 0.2643] [--step-ar 0.2726] [--step-sh 0.2808] [--part-b-us 31.8] [--mb 13.3]"""
import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--step1", type=float, default=0.2643)
ap.add_argument("--step-ar", type=float, default=0.2726)
ap.add_argument("--step-sh", type=float, default=0.2808)
ap.add_argument("--step-classic", type=float, default=0.3086, help="rounds 1-5's sequence on the same box, for the table's first row")
ap.add_argument("--part-b-us", type=float, default=31.8, help="AdamW part B over all touched rows (profiles/r06_dp_one_rank_timeline.txt)")
ap.add_argument("--mb", type=float, default=13.3)
ap.add_argument("--hop-us", type=float, default=4.0)
ap.add_argument("--link-gbs", type=float, default=76.8, help="one direction of one xGMI link")
ap.add_argument("--lat-us", type=float, default=6.0, help="latency of one direct collective")
a = ap.parse_args()


def t_ring_us(n, mb, busbw_gbs, launches=1):
    return 2.0 * (n - 1) / n * mb * 1e6 / (busbw_gbs * 1e9) * 1e6 + 2 * (n - 1) * a.hop_us + (launches - 1) * a.lat_us


def t_direct_us(n, mb):
    return 2.0 * (mb * 1e6 / n) / (a.link_gbs * 1e9) * 1e6 + 2 * a.lat_us


def step_sharded_ms(n):
    return a.step_sh - a.part_b_us * 1e-3 * (1.0 - 1.0 / n)


print("measured on one MI355X: step_1 %.4f ms; one-rank RCCL recording: classic %.4f, all-reduce form %.4f, sharded form %.4f ms "
      "(its AdamW part B: %.1f us over all rows); %.1f MB on the wire in fp32" % (a.step1, a.step_classic, a.step_ar, a.step_sh, a.part_b_us, a.mb))
print("assumed: ring hop %.0f us; direct exchange over %d-1 links at %.1f GB/s per link and direction, %.0f us per collective" % (
    a.hop_us, 8, a.link_gbs, a.lat_us))
print("%-66s %7s %7s %7s" % ("form of the step, wire model", "N = 2", "N = 4", "N = 8"))
rows = []
for bw in (100.0, 150.0, 200.0):
    rows.append(("rounds 1-5 sequence (classic), ring all-reduce at %3.0f GB/s" % bw,
                 [a.step1 / (a.step_classic + t_ring_us(n, a.mb, bw) * 1e-3) for n in (2, 4, 8)], "t_wire(8) = %3.0f us" % t_ring_us(8, a.mb, bw)))
for bw in (100.0, 150.0, 200.0, 300.0):
    rows.append(("all-reduce form (default), ring at %3.0f GB/s bus bandwidth" % bw,
                 [a.step1 / (a.step_ar + t_ring_us(n, a.mb, bw) * 1e-3) for n in (2, 4, 8)], "t_wire(8) = %3.0f us" % t_ring_us(8, a.mb, bw)))
for bw in (150.0, 300.0):
    rows.append(("sharded update, reduce_scatter + all_gather as rings at %3.0f GB/s" % bw,
                 [a.step1 / (step_sharded_ms(n) + t_ring_us(n, a.mb, bw, launches=2) * 1e-3) for n in (2, 4, 8)],
                 "t_wire(8) = %3.0f us" % t_ring_us(8, a.mb, bw, launches=2)))
rows.append(("all-reduce form, direct exchange over all links",
             [a.step1 / (a.step_ar + t_direct_us(n, a.mb) * 1e-3) for n in (2, 4, 8)], "t_wire(8) = %3.0f us" % t_direct_us(8, a.mb)))
rows.append(("sharded update, direct reduce_scatter + all_gather",
             [a.step1 / (step_sharded_ms(n) + t_direct_us(n, a.mb) * 1e-3) for n in (2, 4, 8)], "t_wire(8) = %3.0f us" % t_direct_us(8, a.mb)))
rows.append(("sharded update, direct, 16-bit wire (PVD_DP_WIRE: all-reduce form only today)",
             [a.step1 / (step_sharded_ms(n) + t_direct_us(n, a.mb / 2) * 1e-3) for n in (2, 4, 8)], "t_wire(8) = %3.0f us" % t_direct_us(8, a.mb / 2)))
for label, eff, note in rows:
    print("%-66s %7.2f %7.2f %7.2f   (%s)" % (label, eff[0], eff[1], eff[2], note))
# the bound no schedule of an un-overlapped fp32 exchange of this size beats: all 7 links of a GPU busy in both phases, zero recording tax
floor = 2.0 * (7.0 / 8.0) * a.mb * 1e6 / (7 * a.link_gbs * 1e9) * 1e6
print("bound at N = 8: 2 x 7/8 x %.1f MB over 7 x %.1f GB/s = %.0f us on the wire -> %.2f with no recording tax at all; 0.85 needs <= %.0f us of "
      "wire + tax per step" % (a.mb, a.link_gbs, floor, a.step1 / (a.step1 + floor * 1e-3), (a.step1 / 0.85 - a.step1) * 1e3))
