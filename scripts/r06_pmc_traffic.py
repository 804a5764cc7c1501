"""profiles/pmc_traffic.json from the two counter passes of the default command (scripts/r06_final2.sh): HBM-side bytes per unit of work of the
dominant kernels, FETCH_SIZE (KB) doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE (KB) as reported.
usage: r06_pmc_traffic.py <FETCH summary> <WRITE summary> <pairs per step> <mean iterations> <out.json>"""
import json
import sys


def counters(path, counter):
    out = {}
    for line in open(path):
        parts = line.split()
        if counter in parts:
            i = parts.index(counter)
            out[" ".join(parts[:i])] = (int(parts[i + 1]), float(parts[i + 2]))  # dispatches, sum (KB)
    return out


def main():
    fetch, write = counters(sys.argv[1], "FETCH_SIZE"), counters(sys.argv[2], "WRITE_SIZE")
    pairs, it_mean = int(sys.argv[3]), float(sys.argv[4])

    def find(d, key):
        for k, v in d.items():
            if k.startswith(key):
                return v
        return (0, 0.0)

    res = {"source": "profiles/r06_pmc_FETCH_SIZE.txt and profiles/r06_pmc_WRITE_SIZE.txt: two separate rocprofv3 --kernel-trace --pmc passes (scripts/r06_final2.sh) of the DEFAULT "
                     "workload on the round-6 final tree (python bench.py --steps 1 --warmup 1 --cpu-baseline 0 --no-hints-steps 0); FETCH_SIZE (KB) doubled as MI355X_MICROARCH.md "
                     "prescribes for gfx950 (calibrated for wide coalesced reads; an upper bound for the 16-byte gathers and scalar CSR reads here), WRITE_SIZE (KB) as reported "
                     "(uncalibrated).  Round 5, same method: pair_loop 24.7 MB per pair-iteration, pca_cells 26.7 MB and bsc 79.2 MB per cloud",
           "unit": "bytes per unit of work"}
    nd, fk = find(fetch, "k_pair_loop")
    _, wk = find(write, "k_pair_loop")
    batches = max(1, (nd - 3) // 2)  # two class launches per batch (round 6: the confined class and the other; the re-launch is gone) + three single-pair runs (latency measurement)
    pair_it = batches * pairs * it_mean
    res["pair_loop"] = {"per": "pair_iteration", "bytes": int((2 * fk + wk) * 1024 / pair_it), "dispatches": nd, "batches": batches}
    clouds = batches * pairs * 2 + 768  # + the front-end calibration (64 pairs x 2 passes x 3 shapes)
    for name, kern in (("bsc", "k_fb_bsc"), ("pca_cells", "k_fb_pca_cells")):
        _, f = find(fetch, kern)
        _, w = find(write, kern)
        res[name] = {"per": "cloud", "bytes": int((2 * f + w) * 1024 / clouds)}
    json.dump(res, open(sys.argv[5], "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "source"}))


if __name__ == "__main__":
    main()
