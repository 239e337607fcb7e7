"""Per-family sums of a graph_gaps.txt (scratch/prof_r04.sh): us per step and launches per step."""
import sys, re
fam = [("3x3 halo forward / data gradient (incl. two-source, space-to-depth, folded-shortcut forms)", r"conv_halo"), ("grouped weight gradients", r"conv_wgrad_taps9_group|conv_wgrad_row_group"),
       ("their reduces", r"wgrad_group9_reduce|wgrad_group_reduce"), ("BatchNormalization backward", r"bn_bwd"), ("BatchNormalization forward", r"bn_apply|bn_finalize|bn_partial"),
       ("small-channel forward / data gradient (lean)", r"conv_sc_lean|conv_sc_stream"), ("small-channel weight gradient", r"conv_sc_wgrad"),
       ("stem forward + weight gradient", r"conv_stem"), ("generic per-tap kernel", r"conv_igemm"), ("decoder_stage3_conv1 (scn / scw)", r"conv_scn|conv_scw"),
       ("lone weight gradients + reduces", r"conv_wgrad_dma|conv_wgrad_kernel|wgrad_reduce")]
tot = {k: [0.0, 0.0] for k, _ in fam}; other = [0.0, 0.0]; names = []
for l in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+([0-9.]+) launches/step\s+([0-9.]+) us/step", l)
    if not m: continue
    for k, pat in fam:
        if re.search(pat, m.group(1)):
            tot[k][0] += float(m.group(3)); tot[k][1] += float(m.group(2)); break
    else:
        other[0] += float(m.group(3)); other[1] += float(m.group(2)); names.append((float(m.group(3)), m.group(1).strip()[:40]))
for k, _ in fam: print("%-50s %8.1f us %6.1f launches" % (k, tot[k][0], tot[k][1]))
print("%-50s %8.1f us %6.1f launches   (%s)" % ("everything else", other[0], other[1], ", ".join("%s %.0f" % (n, t) for t, n in sorted(names, reverse=True)[:9])))
print("%-50s %8.1f us" % ("total", sum(v[0] for v in tot.values()) + other[0]))
