"""Test infrastructure only: CPU oracle of the NBMF-MM hot path. Never imported by nbmf_mm_amd/."""
