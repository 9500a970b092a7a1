"""whisper_sae for AMD Instinct MI355X (gfx950): drop-in for the SAE training path of
omarkhursheed/whisper-sae -- ``whisper_sae.sae.model``, ``whisper_sae.sae.training``,
``whisper_sae.config`` and the activation feed -- with the train step running as hand-written HIP
kernels behind the C ABI of ``include/wsae.h`` (``libwsae_hip.so``)."""

__version__ = "0.1.0"
