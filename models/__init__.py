"""Drop-in import path of the reference (`from models.dreamvla_model import DreamVLA`, train.py:16,
eval_calvin.py:10, eval_libero.py:22).  Thin re-exports of dreamvla_amd.*; see INTEGRATION.md."""
