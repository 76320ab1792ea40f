"""CPU oracle for the AutoMoE data-parallel train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`self-driving-model_amd/`) may import,
call, link or execute anything from this directory.  Allowed users: `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` -- and there only as the
checker / the reported CPU baseline, never as the thing measured or shipped.

What is here:
  torch_ref.py   torch-CPU (fp32) restatement of the reference modules on the path:
                 ResNet-18 trunk + the three BDD experts, expert extractors, context extractor,
                 gating network, trajectory policy, AutoMoE composition.
  losses.py      detection set-loss, segmentation CE, gating losses (torch-CPU fp32).
  matcher.py     box ops, GIoU, cost matrix (torch-CPU fp32) + linear-sum-assignment wrapper.
  lsap.c         plain-C restatement of the rectangular LSAP solver scipy 1.15.3 ships
                 (modified Jonker-Volgenant, Crouse 2016) -- the algorithm the HIP kernel follows.
  Makefile       builds oracle/_ref/liblsap_oracle.so from lsap.c.

Pinning status (see DESIGN.md "Oracle"):
  * gating / policy / context / extractors: pinned -- checked against the importable reference
    modules in this container by tests/golden/make_golden.py, vectors committed in tests/golden/.
  * LSAP: pinned against scipy.optimize.linear_sum_assignment 1.15.3 (the reference's call,
    training/hungarian_matcher.py:79) on committed vectors and live at test time.
  * ResNet-18 trunk, box_convert / generalized_box_iou: torchvision is absent from the image, the
    reference's tests hold no numeric vectors for them -> "parity unpinned": restated from the
    published architecture / formulas, checked structurally (param counts, state_dict keys, shapes).
"""
