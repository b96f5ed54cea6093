"""causal-gen on MI355X: the HVAE image-mechanism / counterfactual hot path of biomedia-mira/causal-gen behind
the reference's own Python surface (``vae.HVAE``, ``dmol.DmolNet``, ``dscm.DSCM``), computed by hand-written
gfx950 HIP kernels in ``libcgen_hip.so`` (C ABI: include/cgen_hip.h).  There is no CPU or ATen fallback: the
model classes raise if the library or a GPU is missing."""
__version__ = "0.1.0"
