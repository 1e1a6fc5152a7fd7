"""Which parity cases / chunkings make k_assocb stop for reason 6 (AB_BAIL_REACH: a candidate from a column older than the first unpublished one)?
usage: python tools/find_reach_bail.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cases, util
found = []
for name in ["x_s64_refused_attach"] + [n for n in cases.ALL_CASES if n != "x_s64_refused_attach"]:
    stream, cfg, tf = cases.build_case(name)
    for chunks in ([stream.sensor.num_columns], [97, 1, 200], [61], [131, 7], [33], [250, 19]):
        box = {}
        try:
            util.run_and_compare(stream, cfg, chunks=chunks, robot_tf=tf, engine_setup=lambda e: box.__setitem__("e", e))
            why = box["e"].batch_counters()["bail_reasons"]
            if sum(why) > 0:
                print(name, chunks, "reasons", why[:7], flush=True)
            if why[6] > 0:
                found.append((name, chunks))
        except AssertionError as ex:
            print(name, chunks, "MISMATCH", str(ex)[:200], flush=True)
print("found", found)
