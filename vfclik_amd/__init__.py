"""vfclik_amd -- MI355X-native batched replacement for vfclik's per-cycle control loop.

Hot path (HIP, gfx950): ``vfclik_amd/csrc/vfik_kernel.hip`` (kernels) + ``vfik_abi.cpp`` (host side) behind the C-ABI of ``include/vfik.h``.
Host side (this package) mirrors the reference's port / handler interface:

    engine          ctypes binding of the C-ABI (fails loudly without the HIP library)
    chain, robots   kinematic chain descriptions (what the reference hides in ``Lafik(config)``)
    fields          /param message handling -> field records (scripts/vf:209-293)
    ports           in-process stand-in for the YARP port/bottle surface the reference uses
    handlers        src/handlers.py API, batched
    command_mixer   src/command_mixer.py API, batched
    vf_module       the per-cycle modules (vf, nullspace, debug_jointlimits, bridge mixer) for B arms behind the reference's ports
    object_feeder   scripts/object_feeder translation layer (goal, goalAndNormal, obstacles -> /param records)
    sharding        a global batch over the GPUs of a node (ShardedEngine), launcher: one process per GPU
"""
__version__ = "0.3.0"
