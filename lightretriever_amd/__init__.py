"""lightretriever_amd: MI355X-native corpus-embedding + flat-IP search path of LightRetriever."""
from .encoder import EncoderConfig, LrxEncoder, interleave_gate_up, lora_merge, rope_tables  # noqa: F401
from .index import FlatIPIndex, merge_topk  # noqa: F401
