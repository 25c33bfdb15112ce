"""Drop-in surface with the reference's module names (`SOC_matching.utils`,
`.method`, `.models`, `.experiment_settings.*`), backed by `socmx`."""
