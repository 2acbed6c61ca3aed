"""Host-side mirror of the reference's ``RegressionNetwork/`` package (same module names)."""
