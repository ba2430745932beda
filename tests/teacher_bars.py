"""the bars of the teacher-forced GPU tests (importable without a GPU: tests/test_teacher_harness.py checks that a wrong
state trips them).  Documented in tests/test_gpu_teacher.py."""
BARS = dict(loss_first=2e-5, loss_rel=2e-4, frac=1e-3, max_klr=2.02, med_klr=0.002, m_rel=0.1, v_rel=5e-3)
