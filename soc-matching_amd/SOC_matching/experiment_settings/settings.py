"""Reference-named `define_variables` (experiment_settings/settings.py:210-301)."""
from socmx.settings import define_variables  # noqa: F401
