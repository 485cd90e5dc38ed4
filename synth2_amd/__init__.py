"""synth2_amd — MI355X-native voice-render path behind s2_lib's buffer-fill API.

csrc/ holds the HIP kernels and the C ABI (libs2r.so, declared in include/s2r.h);
synth.py is the ctypes mirror of the reference's ``Synth`` used by tests and bench.py.
"""
from .synth import (Adsr, Note, Patch, S2rError, SampleRateKhz, Synth, Velocity, VoicePool,  # noqa: F401
                    default_patch, load_library, parse_patch, shard_pool_indices, stream_frame_json, sum_partials_device,
                    OSC_SAW, OSC_SINE, OSC_SQUARE, OSC_TRIANGLE, OSC_DPW_SAW, OSC_DPW_SQUARE, OSC_DPW_TRIANGLE, VOICE_STATE_DTYPE, NOTE_EVENT_DTYPE, LAYER_CALL_DTYPE,
                    FILT_ONEPOLE, FILT_LP1, FILT_HP1, FILT_LP2, FILT_HP2, FILT_BP2,
                    FILT_SVF_LP, FILT_SVF_BP, FILT_SVF_HP)
