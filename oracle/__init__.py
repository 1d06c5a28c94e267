"""CPU oracle for the FEM hot path — TEST INFRASTRUCTURE ONLY (see fem_oracle.h)."""
