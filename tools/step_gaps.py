"""Where one step's wall time goes between kernels (profiles/r06_experiments/06_step_timeline.md).

Input: the kernel trace of an UN-instrumented bench job, e.g. on the GPU box
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o g -- python bench.py --side --no-prof --no-parity \
        --no-cpu-baseline --no-realistic --no-fp32-side --steps 6 --warmup 2
    python tools/step_gaps.py out/g_kernel_trace.csv
Prints (1) per step: span, time with at least one kernel resident, idle time, the largest gaps and what follows them; (2) the step's
phases: the small-kernel prologue (sampler, batch assembly, packing), the main part (first to last large kernel), the tail before the
optimizer (embedding backward, gradient norms) and the two AdamW launches.  A step is delimited by its adamw_kernel launches (two
per step: one per tower)."""
import collections
import csv
import statistics
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    return sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)


def gaps(iv, nsteps=4, per=2):
    ad = [i for i, (s, e, n) in enumerate(iv) if n.startswith("adamw")]
    a, b = ad[-1 - per * nsteps], ad[-1]
    t0, t1 = iv[a][1], iv[b][1]
    busy, cur_s, cur_e, out, prev = 0, None, None, [], None
    for s, e, n in iv[a + 1:b + 1]:
        if cur_e is None:
            cur_s, cur_e, prev = s, e, n
            continue
        if s > cur_e:
            out.append((s - cur_e, prev, n))
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = n
    busy += cur_e - cur_s
    print("per step: span %.2f ms, >= 1 kernel resident %.2f ms, idle %.2f ms; %d launches, %d gaps" %
          ((t1 - t0) / 1e6 / nsteps, busy / 1e6 / nsteps, (t1 - t0 - busy) / 1e6 / nsteps, (b - a) / nsteps, len(out) / nsteps))
    out.sort(reverse=True)
    for g, p, n in out[:10]:
        print("  %8.1f us after %-48s before %s" % (g / 1e3, p[:48], n[:48]))
    tot, cnt = collections.Counter(), collections.Counter()
    for g, p, n in out:
        tot[n[:60]] += g
        cnt[n[:60]] += 1
    print("idle in front of a kernel: ms / step, gaps / step, mean us")
    for n, v in tot.most_common(8):
        print("  %-60s %6.2f %6.1f %7.1f" % (n, v / 1e6 / nsteps, cnt[n] / nsteps, v / cnt[n] / 1e3))
    print("median gap %.1f us" % (statistics.median(g for g, _, _ in out) / 1e3))


def phases(iv):
    ad = [i for i, (s, e, n) in enumerate(iv) if n.startswith("adamw")]

    def big(n):
        return any(k in n for k in ("gemm_nt_p3", "gemm_nt_p5", "gemm_tn5", "gemm_nt_xp", "gemm_tn_xq", "mha_", "ln_fwd_kernel", "ln_bwd_kernel"))
    for k in range(len(ad) - 8, len(ad) - 2, 2):
        a, b = ad[k + 1], ad[k + 2]
        seg = iv[a + 1:b]
        t0 = iv[a][1]
        fb = next(i for i, x in enumerate(seg) if big(x[2]))
        lb = max(i for i, x in enumerate(seg) if big(x[2]))
        print("prologue %.2f ms (%d kernels) | first to last large kernel %.2f ms | tail before AdamW %.2f ms (%d kernels) | AdamW x 2 %.2f ms" %
              ((seg[fb][0] - t0) / 1e6, fb, (seg[lb][1] - seg[fb][0]) / 1e6, (iv[b][0] - seg[lb][1]) / 1e6, len(seg) - 1 - lb, (iv[b + 1][1] - iv[b][0]) / 1e6))
        if k == len(ad) - 4:
            print("  tail kernels:")
            for s, e, n in seg[lb + 1:]:
                print("    %8.1f us  +%8.1f  %s" % ((e - s) / 1e3, (s - seg[lb][1]) / 1e3, n[:90]))


if __name__ == "__main__":
    iv = load(sys.argv[1])
    gaps(iv)
    phases(iv)
