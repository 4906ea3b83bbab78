"""dff/utils.py of the reference: `select_focus_dist` (dff/utils.py:4-50)."""
from aadff.focal_stack import select_focus_dist   # noqa: F401
