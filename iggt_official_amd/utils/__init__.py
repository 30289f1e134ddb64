"""The callers' side of the forward path, on the GPU: image loading / preprocessing (load_fn), pose decoding (pose_enc),
depth unprojection (geometry) and checkpoint ingestion (model).  Mirrors reference iggt/utils/{load_fn,pose_enc,geometry}.py
and utils/model.py for the functions demo.py uses (demo.py:36-38,51); `iggt/utils/*.py` and `utils/model.py` at the
repository root re-export them under the reference's import paths."""
