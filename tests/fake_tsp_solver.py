#!/usr/bin/env python3
"""Stand-in for the Lazy solver's external TSP binary (reference src/lazy.h:93-99 runs
`<path> --map-type=TSP_FILE --use-path-files-folder=false --use-prm=false --tsp-solver=<type> --problem=<file>`; the
real `obst_tsp` is not public).  Reads the TSPLIB LOWER_DIAG_ROW matrix, finds the shortest closed tour from city 0 by
brute force (ties: the lexicographically smallest order) and writes the one result line the solver parses
(src/lazy.h:286-300): "<length> , <c0> , <c1> , ... , <c0>" into <id>tempTsp.result beside the problem file."""
import itertools
import os
import sys


def read_matrix(path):
    lines = open(path).read().split("\n")
    n = int([l for l in lines if l.startswith("DIMENSION")][0].split(":")[1])
    rows = lines[lines.index("EDGE_WEIGHT_SECTION") + 1:]
    d = [[0.0] * n for _ in range(n)]
    for i in range(n):
        vals = rows[i].split()
        for j in range(i):
            d[i][j] = d[j][i] = float(vals[j])
    return d


def best_tour(d):
    n = len(d)
    best, order = None, None
    for perm in itertools.permutations(range(1, n)):
        t = (0,) + perm + (0,)
        length = sum(d[t[k]][t[k + 1]] for k in range(n))
        if best is None or length < best:
            best, order = length, t
    return best, order


def main():
    problem = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--problem=")][0]
    length, order = best_tour(read_matrix(problem))
    out = os.path.join(os.path.dirname(problem), os.path.basename(problem).replace("tempTsp.tsp", "tempTsp.result"))
    with open(out, "w") as f:
        f.write(" , ".join([repr(length)] + [str(c) for c in order]) + "\n")


if __name__ == "__main__":
    main()
