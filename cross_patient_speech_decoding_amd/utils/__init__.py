"""Synthetic-data helpers shared by the benchmarks, the CLI and the tests."""
