"""Drop-in import alias: `from iggt.models.vggt import IGGT` (what the reference's demo.py:35 does)
resolves to the MI355X implementation in `iggt_official_amd`.  Thin re-exports only."""
