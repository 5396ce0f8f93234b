"""Code-generation guards for the kernels the bench line depends on (CPU: disassembles libmmcmc.so's gfx950 code objects
with llvm-objdump, no GPU needed).  Two properties of the machine code are not properties of the source and were seen to
flip with harmless-looking edits in round 4:

* whether the compiler unrolls the transition wave's batch loop of the HMC split kernel (run-time trip count <= 8).  With a
  slightly smaller transition body it stopped unrolling the collecting phase and the kernel lost 8 % (0.185 -> 0.197 ms) with
  every test green;
* whether a lane-state struct stays in registers: `c ? a.w[i] : b.w[i]` on two array lvalues selects the ADDRESS and sends
  the struct to scratch memory (config 5: 466 -> 654 ms), again with every test green.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
SO = os.path.join(ROOT, "mini_mcmc_amd", "libmmcmc.so")

HMC = "mm_run_split_kernelIf9mm_targetIfLi4ELi3EELi1ELi10E"  # <float, RosenbrockND<3>, HMC, L = 10, ...>
MH = "mm_run_split_kernelIf9mm_targetIfLi0ELi2EELi0ELi0E"    # <float, Gaussian2D, MH, ...>
LGQ = "mm_nuts_lgq_kernelILi32ELi1E"                          # config 5's persistent scheduler, one wave per SIMD
PAIR = "mm_nuts_pair_kernelIfd9mm_targetIfLi4ELi3EELb1E"      # small-D NUTS (RosenbrockND(3), mode 0): the tick's draws once lived in scratch


def _text(want):
    import isa_mix

    if not (os.path.exists(SO) and os.path.exists(isa_mix.OBJDUMP)):
        pytest.skip("libmmcmc.so or llvm-objdump not available")
    name, lines = isa_mix.kernel_text(SO, want)
    assert name, f"no kernel matching {want} in libmmcmc.so"
    return [ln.split("//")[0].strip() for ln in lines]


@pytest.mark.parametrize("want", [HMC, MH, LGQ, PAIR])
def test_hot_kernels_keep_their_state_in_registers(want):
    text = _text(want)
    spills = [ln for ln in text if ln.startswith("scratch_") or ln.startswith("buffer_load_dword v") and "offen" in ln]
    assert not spills, f"{want}: {len(spills)} scratch accesses, e.g. {spills[:3]}"


def test_hmc_split_kernel_unrolls_both_phases_of_the_transition_wave():
    text = _text(HMC)
    # a transition = 10 leapfrog steps of 6 packed instructions (+ the energies): ~62 packed instructions; the ring half is
    # 8 transitions, unrolled once for the burn-in and once for the collecting phase -> ~1000; one phase rolled -> ~520
    n_pk = sum(1 for ln in text if ln.startswith("v_pk_"))
    assert n_pk >= 900, f"{n_pk} packed instructions: a phase of the transition wave is no longer unrolled"
