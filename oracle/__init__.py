"""CPU oracle for the inquiSTR `call` hot path: TEST INFRASTRUCTURE ONLY (see inq_oracle.h)."""
