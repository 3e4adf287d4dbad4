from dreamvla_amd.action_model.models import (ActionEmbedder, DiT, DiTBlock, FinalLayer, HistoryEmbedder,  # noqa: F401
                                              LabelEmbedder, TimestepEmbedder)
