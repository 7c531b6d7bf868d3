//! deflate_bytes_conf / _zlib_conf / _gzip_conf (src/lib.rs:137-286) of the real crate for every file under
//! the fixture directory at every named level, written as <out>/<relative path>.<level>.<raw|zlib|gzip>.
use deflate::{deflate_bytes_conf, deflate_bytes_gzip_conf, deflate_bytes_zlib_conf, Compression, CompressionOptions};
use gzip_header::GzBuilder;
use std::{env, fs, path::Path};

fn levels() -> Vec<(&'static str, CompressionOptions)> {
    vec![
        ("fast", Compression::Fast.into()),
        ("default", Compression::Default.into()),
        ("best", Compression::Best.into()),
        ("rle", CompressionOptions::rle()),
        ("huffman_only", CompressionOptions::huffman_only()),
    ]
}

fn walk(dir: &Path, base: &Path, out: &Path) {
    for e in fs::read_dir(dir).unwrap() {
        let p = e.unwrap().path();
        if p.is_dir() {
            walk(&p, base, out);
            continue;
        }
        let data = fs::read(&p).unwrap();
        let rel = p.strip_prefix(base).unwrap();
        for (name, o) in levels() {
            let stem = out.join(rel);
            fs::create_dir_all(stem.parent().unwrap()).unwrap();
            let f = |ext: &str| format!("{}.{}.{}", stem.display(), name, ext);
            fs::write(f("raw"), deflate_bytes_conf(&data, o)).unwrap();
            fs::write(f("zlib"), deflate_bytes_zlib_conf(&data, o)).unwrap();
            fs::write(f("gzip"), deflate_bytes_gzip_conf(&data, o, GzBuilder::new())).unwrap();
        }
    }
}

fn main() {
    let a: Vec<String> = env::args().collect();
    let (src, dst) = (Path::new(&a[1]), Path::new(&a[2]));
    walk(src, src, dst);
}
