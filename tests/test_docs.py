"""The integration notes quote the Rust binding under ffi/ instead of paraphrasing it: every ```rust block of their
section on the Rust side must be found word for word in ffi/src/ffi.rs or ffi/src/flatten.rs (two documents, one
binding)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_quotes_the_rust_binding_verbatim():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 2. The Rust side"):text.index("## 3.")]
    blocks = re.findall(r"```rust\n(.*?)```", section, flags=re.S)
    assert len(blocks) >= 2
    sources = [open(os.path.join(ROOT, "ffi", "src", name)).read() for name in ("ffi.rs", "flatten.rs")]
    for block in blocks:
        # a block may quote two adjacent pieces of one file with a blank line between them
        for piece in (p for p in block.split("\n\n") if p.strip()):
            assert any(piece.strip("\n") in src for src in sources), piece[:200]
    # the signature the reference's callers see is kept to the letter (src/renderer.rs:9-18)
    flat = sources[1]
    assert "pub fn render_world<'a>(background: &Vec3, world: &'a BoundingVolumeHierarchy, camera: &Camera, screen_width: usize," in flat
    assert "rand::random()" in flat and "side_material" not in flat
