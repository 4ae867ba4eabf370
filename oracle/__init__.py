"""CPU oracle for the MPPI.forward() hot path — TEST INFRASTRUCTURE, never imported by the product."""
