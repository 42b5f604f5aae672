"""CPU oracle for the aligned-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker /
the timed CPU baseline.  The product (``cross_patient_speech_decoding_amd``)
never imports this package and raises if its HIP library is missing.

Modules
-------
align_oracle    numpy float64 restatement of alignment/{alignment_utils,AlignCCA,
                JointPCA}.py and of process_aligner (datamodules.py:515-574);
                pinned by tests/golden/align_*.npz generated from the reference.
mcca_oracle     restatement of alignment/AlignMCCA.py around a from-the-paper
                regularised MCCA (mvlearn is absent: PARITY UNPINNED for the
                third-party call, see module header).
seq2seq_oracle  plain torch.nn CPU restatement of nn_models/models.py
                (TemporalConv, EncoderRNN, DecoderRNN, Seq2SeqRNN, cmat_acc) and
                of the Lightning optimisation recipe; pinned by
                tests/golden/seq2seq_*.npz generated from the reference.
"""
