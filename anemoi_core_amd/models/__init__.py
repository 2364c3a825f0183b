from .encoder_processor_decoder import AnemoiModelEncProcDec  # noqa: F401
