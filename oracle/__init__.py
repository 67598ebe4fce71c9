"""CPU oracle for the Tacotron 2 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``tacotron2_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the checker
(or as the timed CPU baseline), never as the thing shipped.

Parity status
-------------
* ``tacotron2_ref`` (encoder, conditioning, prenet, decoder step, attention, loop,
  postnet, masking, loss): PINNED against the reference itself.  The reference's
  ``model.tacotron2.Tacotron2`` was imported in the build container by
  ``oracle/make_golden.py`` to emit ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
  checks this restatement against those vectors.
* ``logmel_ref`` (STFT + slaney mel + log): PARITY UNPINNED.  The arithmetic lives in the
  un-vendored third-party package ``speech_utils`` (reference ``requirements.txt:12``, a git
  URL with no pinned version) which is absent here and whose outputs no reference file holds.
  The definition is restated from in-repo evidence (``datasets/prosody_dataset.py:39-50,67``,
  ``run/say.py:161-171``).
"""
