"""Lower a bench workload on the CPU box into gpurun_in/cache (travels with gpurun), so that GPU minutes are not spent
on Python lowering:  python tools/prebuild_cache.py <workload> [batch]   (same cache key bench.py computes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                          # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sha256_2048"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_BATCH.get(name, 4096)
root = os.path.join(str(bench.ROOT), "gpurun_in", "cache")
os.makedirs(root, exist_ok=True)
cp, s, _ = bench.get_compiled(name, batch, root, 0, None)
print("cached", cp.dir, "%.1f s" % s)

# big artefacts are gzipped for the snapshot (bench.get_compiled unpacks them on the GPU box)
import glob, gzip, shutil
total = sum(os.path.getsize(f) for f in glob.glob(os.path.join(cp.dir, name + ".*")) if not f.endswith(".gz"))
if total > (200 << 20):
    for ext in (".cwt", ".dat", ".r1cs"):
        f = os.path.join(cp.dir, name + ext)
        with open(f, "rb") as fi, gzip.open(f + ".gz", "wb", compresslevel=1) as fo:
            shutil.copyfileobj(fi, fo, 1 << 24)
        os.unlink(f)
    print("gzipped", total >> 20, "MB")
