"""CPU oracle of the ray-casting path: TEST INFRASTRUCTURE ONLY (see rc_oracle.h)."""
