// kmer_probe — prints what kmerutils 0.0.14 itself yields for the SURVEY Appendix-B sequence, and where
// hyperminhash 0.1.4 puts the first k-mer, so that an HMH image mismatch can be split between U5 (k-mer value)
// and U1 (which xxh3_128 half is x).  Compare with the table in SURVEY.md Appendix B.
// Written without a Rust toolchain at hand (none in the build image): if an import path is off, fix it here —
// nothing else in the repository depends on this program.
use hyperminhash::Sketch;
use kmerutils::base::kmergenerator::{KmerSeqIterator, KmerSeqIteratorT};
use kmerutils::base::sequence::Sequence;
use kmerutils::base::{CompressedKmerT, Kmer16b32bit, Kmer64bit, KmerT};
use xxhash_rust::xxh3::{xxh3_128_with_seed, xxh3_64_with_seed};

fn main() {
    let text = b"ACGTTGCATGCATCGATCGGATTACA";
    let seq = Sequence::new(text, 2);

    println!("k=16 (Kmer16b32bit): position forward reverse_complement canonical");
    let mut it16 = KmerSeqIterator::<Kmer16b32bit>::new(16, &seq);
    let mut pos = 0;
    let mut first16: u32 = 0;
    while let Some(km) = it16.next() {
        let rc = km.reverse_complement();
        let canon = km.min(rc);
        if pos == 0 { first16 = canon.get_compressed_value(); }
        println!("{} {:#x} {:#x} {:#x}", pos, km.get_compressed_value(), rc.get_compressed_value(), canon.get_compressed_value());
        pos += 1;
    }

    println!("k=21 (Kmer64bit): position forward reverse_complement canonical");
    let mut it21 = KmerSeqIterator::<Kmer64bit>::new(21, &seq);
    pos = 0;
    while let Some(km) = it21.next() {
        let rc = km.reverse_complement();
        let canon = km.min(rc);
        println!("{} {:#x} {:#x} {:#x}", pos, km.get_compressed_value(), rc.get_compressed_value(), canon.get_compressed_value());
        pos += 1;
    }

    // the bytes lash hashes for HyperMinHash (the low 32 bits, little-endian) and both hashes of them, seed 42
    let bytes = first16.to_le_bytes();
    let h128 = xxh3_128_with_seed(&bytes, 42);
    println!("xxh3_128(le4 of first canonical 16-mer, 42) = {:#034x}  (high64 {:#x}, low64 {:#x})", h128, (h128 >> 64) as u64, h128 as u64);
    println!("xxh3_64(le8 of the same value, 42) = {:#x}", xxh3_64_with_seed(&(first16 as u64).to_le_bytes(), 42));

    // one k-mer into an empty HyperMinHash: which register becomes non-zero tells U1
    // (14494 = 0x1682 if x is the high half, 14219 = 0x0bea if x is the low half)
    let mut sk = Sketch::default();
    sk.add_bytes_with_seed(&bytes, 42);
    let mut image: Vec<u8> = Vec::new();
    sk.save(&mut image).expect("save");
    println!("hyperminhash image: {} bytes", image.len());
    for (i, pair) in image.chunks(2).enumerate() {
        if pair[0] != 0 || pair[1] != 0 {
            println!("non-zero register bytes at byte offset {}: {:02x} {:02x}", 2 * i, pair[0], pair[1]);
        }
    }
}
