import sys, numpy as np
n_rows, n_cols, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(1)
vals = (rng.integers(8, 40, (n_rows, 1)) + rng.integers(-2, 3, (n_rows, n_cols))).astype(np.float32)
hit = rng.integers(0, n_cols, n_rows)
vals[np.arange(n_rows), hit] *= rng.choice([1.0, 1.0, 6.0], n_rows).astype(np.float32)
half = rng.random((n_rows, n_cols)) < 0.2
with open(path, "w") as f:
    f.write("chromosome\tbegin\tend\t" + "\t".join(f"S{k // 2}_H{1 + k % 2}" for k in range(n_cols)) + "\n")
    for i in range(n_rows):
        cells = [("NaN" if (i + k) % 37 == 0 else (f"{int(v)}.5" if h else str(int(v)))) for k, (v, h) in enumerate(zip(vals[i], half[i]))]
        f.write(f"chr{1 + i % 22}\t{1000 * i}\t{1000 * i + 60}\t" + "\t".join(cells) + "\n")
