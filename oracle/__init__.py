"""CPU oracle for the SFR-on unlearning hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline.  The product path (the package ``unified-unlearning-w-remain-geometry_amd``,
imported as ``sfron``) never routes through this package and fails loudly when
its HIP library is missing.

The oracle is a plain-PyTorch fp32 restatement (CPU) of the reference's
algorithm for the path named in BASELINE.json; every function cites the
reference file:line it follows.  Pinning status:

* diffusion loss / q_sample / vb terms   -- pinned by golden vectors generated
  from the imported reference (tests/golden/make_golden.py).
* DiT module arithmetic in DiT/models.py  -- pinned by golden vectors, EXCEPT
  the three timm classes (PatchEmbed, Attention, Mlp): timm is an un-vendored,
  un-pinned dependency of the reference (DiT/environment.yml lists bare
  ``timm``) and is absent here, so that boundary is "parity unpinned" and
  follows timm's published behaviour.
* Adam / clip_grad_norm_ / LayerNorm / GELU -- torch itself is the reference.
* DDPM loss, adaptive loss, EMAHelper, mask arithmetic -- pinned by golden
  vectors from the imported reference.
* DDPM Conditional_Model (ddpm_ref.py): state_dict keys, parameter count,
  seeded forward (train / test), gradients and a 2-iteration SFR-on
  trajectory -- pinned by golden vectors from the imported reference class.
* sampling (space_timesteps, respaced tables, p_sample, p_sample_loop,
  forward_with_cfg) -- pinned by golden vectors from the imported reference.
"""
