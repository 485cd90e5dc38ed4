"""CPU oracle — TEST INFRASTRUCTURE ONLY (see s2_oracle.h)."""
