"""Which chunkings of tests/cases.py: x_s64_refused_attach make k_assocb stop for reason 4 (AB_BAIL_DEAD: a parent chain that ends in a tree finished
in an EARLIER group)? A call boundary right in front of the second post of a pair puts the finish and the refused attach into different launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cases, util
stream, cfg, tf = cases.build_case("x_s64_refused_attach")
k, i, found = 60, 0, []
gaps = [20, 20, 19, 20, 18, 20, 20, 17, 20, 20, 20, 16, 20, 20]
while k + 25 < stream.n_firings and len(found) < 4:
    gap = gaps[i % len(gaps)]
    b = k + gap
    for first in (b, b + 1, b + 2):
        box = {}
        try:
            util.run_and_compare(stream, cfg, chunks=[first, 100000], robot_tf=tf, engine_setup=lambda e: box.__setitem__("e", e))
            why = box["e"].batch_counters()["bail_reasons"]
            print("pair", (k, b), "first call", first, "reasons", why[:7], flush=True)
            if why[4] > 0:
                found.append(first)
        except AssertionError as ex:
            print("pair", (k, b), "first call", first, "MISMATCH", str(ex)[:200])
    k += gap + 31 + (i % 5)
    i += 1
print("found", found)
