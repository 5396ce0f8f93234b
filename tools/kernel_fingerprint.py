"""sha256 over the sources that define the sampling kernels (csrc headers of the split-role / one-wave kernels, the
samplers, the stream, the targets, the math): tools/summarize_pmc.py stores it with every counter summary, bench.py
recomputes it and quotes a summary only while it still describes the kernel that is being timed."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLING_KERNEL_SOURCES = ("mm_split_kernels.h", "mm_kernels.h", "mm_samplers.h", "mm_rng.h", "mm_icdf_table.h", "mm_targets.h",
                           "mm_math.h", "mm_params.h", "mm_inst.inc")


def sampling_kernel_sources_sha256() -> str:
    h = hashlib.sha256()
    for f in SAMPLING_KERNEL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, "mini_mcmc_amd", "csrc", f), "rb").read())
    # the compile flags, not the list of translation units
    for line in open(os.path.join(ROOT, "mini_mcmc_amd", "csrc", "Makefile")):
        if line.startswith(("HIPFLAGS", "ARCH")):
            h.update(line.encode())
    return h.hexdigest()


if __name__ == "__main__":
    print(sampling_kernel_sources_sha256())
