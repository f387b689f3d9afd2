"""Zero-edit drop-in modules: put THIS directory on ``sys.path`` in front of the reference's ``code/LJ`` / ``code/water``
and the rollout drivers run unchanged —

    from train_network_lj import ParticleNetLightning              # code/LJ/test_script/test_langevin.py:58
    from train_network_tip3p import ParticleNetLightning           # code/water/test_script/test_nosehoover.py:64
    from train_network_real_large import ParticleNetLightning      # code/water/test_script/test_nosehoover_hb.py:64

    model = ParticleNetLightning(args).load_from_checkpoint(PATH, args=args)
    model.load_training_stats(SCALER_CKPT); model.cuda(); model.eval()
    force = model.predict_forces(pos)

Each module is named like the reference's, exports the names the reference's module exports on the inference path
(``ParticleNetLightning``, ``build_model``, ``BOX_SIZE``, ``CUTOFF_RADIUS``, ``NUM_OF_ATOMS``, ``create_water_bond``) with
the reference's values and constructor signature, and reads its module constants when a wrapper is constructed — editing
``NUM_OF_ATOMS`` / ``BOX_SIZE`` / ``CUTOFF_RADIUS`` in the module (how the reference is re-targeted, e.g. the commented-out
TIP4P lines of train_network_tip3p.py:31-32) works the same way.  Written from scratch on gamd_amd.compat; nothing of the
training half (training_step, dataloaders, optimisers: SURVEY.md section 8 "not planned") exists here, and those methods
raise NotImplementedError naming this scope.
"""
