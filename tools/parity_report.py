"""tools/parity_report.py AFTER.json [BEFORE.json] > profiles/rNN_parity_report.txt -- the whole-rig parity numbers of tools/flip_kinds.py
(oracle/rig_parity.py: the HIP path against the reference build on all 160 cameras of config C, forward + backward) as the text report
kept under profiles/.  BEFORE = the same run under a library built from an earlier commit (GSR_LIB_PATH)."""
import json, sys


def block(title, d):
    s = d["summary"]
    keep = {k: s[k] for k in ("views", "flips_per_view", "flips_per_view_by_tensor_max", "largest_image_flip", "largest_gradient_flip",
                              "radii_diff_max", "input_bits_mean") if k in s}
    fk = s.get("flip_kinds", {})
    keep["flip_kinds"] = {k: fk[k] for k in ("elements_by_kind", "share_by_kind", "per_view_mean_elements", "thresholds", "what") if k in fk}
    print("# " + title)
    print(json.dumps(keep, indent=1))


def share(d, pred):
    e = d["summary"].get("flip_kinds", {}).get("elements_by_kind", {})
    tot = sum(e.values())
    return (sum(v for k, v in e.items() if pred(k)), tot)


after = json.load(open(sys.argv[1]))
before = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
print("# whole-rig parity against the reference build (oracle/_ref = the reference's own kernels compiled for gfx950): tools/flip_kinds.py 1,")
print("# all 160 cameras of config C, forward + backward, one MI355X.  'flips' = elements outside tests/parity.py's tolerances, per view summed")
print("# over the image and six gradient tensors; every flagged element is attributed to the hard decision (alpha >= 1/255, T < 1e-4) that the")
print("# two rasterizers took differently: *_inputs = the per-Gaussian projected centre / conic differ in their last bits and decide the pair,")
print("# *_eval = identical inputs, the margin lies within the rounding of exp / of the product.")
if before is not None:
    a, t = share(before, lambda k: k.endswith("_inputs"))
    u, _ = share(before, lambda k: k == "unexplained")
    print("# BEFORE (commit 4ffd22c: straightforward projection formulas): %d flagged elements, %d attributed to a decision (the classifier walks at"
          % (t, t - u))
    print("#   most 600 Gaussians per view); of the attributed ones %d (%.0f %%) come from differing input bits"
          % (a, 100.0 * a / max(t - u, 1)))
    c, t2 = share(before, lambda k: "T_cut" in k)
    print("#   T-cut decisions: %d of %d (%.1f %%); alpha-cut: the rest" % (c, t2, 100.0 * c / max(t2, 1)))
a, t = share(after, lambda k: "T_cut" in k)
print("# AFTER (gsr_ref_order.h: projection in the reference build's operation order): means2D / conic / radii bit-identical on every view;")
print("#   %d flagged elements are left on the whole rig, %d of them (%.1f %%) T-cut, the others alpha-cut by evaluation (exp2 vs expf)" % (t, a, 100.0 * a / max(t, 1)))
if before is not None:
    block("BEFORE", before)
block("AFTER", after)
print("# AFTER, per view: view, total flips, image flips, largest image deviation, largest gradient deviation (relative to the tensor's max)")
for r in after["per_view"]:
    tot = sum(r["flips"].values())
    print("%4d %5d %4d %.2e %.2e" % (r["view"], tot, r["flips"]["color"], r["worst"]["color"],
                                     max(v for k, v in r["worst"].items() if k != "color")))
if before is not None:
    print("# BEFORE, per view: view, total flips, image flips")
    for r in before["per_view"]:
        print("%4d %5d %4d" % (r["view"], sum(r["flips"].values()), r["flips"]["color"]))
