"""Drop-in import surface: the reference's module paths (``src.utils.unets`` ...) re-exporting the MI355X-native
implementations in ``microbeseg_amd`` so that its GUI / scripts keep working unchanged."""
